"""The multi-GPU path on device code (include/o2v_hip.h, multi-GPU section), on whatever GPUs the
box has: with one GPU the ranks share it (the collectives then run over host memory: RCCL refuses two ranks on one
device), with several the same tests use one GPU per rank and RCCL over xGMI.

Bar: the union of the ranks' slabs is bit-identical to the single-GPU result and to the oracle's (the reference voxelizes
independent chunks the same way, src/obj2voxel.cpp:226-243, src/voxelization.cpp:440-444)."""
import ctypes as C
import os
import socket
import sys

import numpy as np
import pytest
import torch  # noqa: F401  first: torch bundles its own HIP runtime and RCCL, which must be the copies this process loads

from obj2voxel_amd import meshes

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _devices(n):
    from obj2voxel_amd import hip
    have = hip.device_count()
    return list(range(n)) if have >= n else [r % have for r in range(n)]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _textured_mesh():
    from obj2voxel_amd import hip
    v, uv = meshes.uv_sphere(60, with_uv=True)
    big = np.array([[-0.9, -0.8, -0.7, 0.9, -0.6, 0.2, 0.1, 0.9, 0.8]], np.float32)   # subdivided, spans every slab
    v = np.concatenate([v, big])
    uv = np.concatenate([uv, np.array([[0, 0, 1, 0, 0.5, 1]], np.float32)])
    T = len(v)
    types = np.full(T, hip.TRI_TEXTURED, np.uint32)
    types[::5] = hip.TRI_UNTEXTURED
    return v, dict(uvs=uv, types=types, colors=meshes.triangle_colors(T), texids=np.zeros(T, np.int32)), \
        [(meshes.checker_texture(64, 8), 1)]


@pytest.mark.parametrize("n_ranks", [2, 3])
@pytest.mark.parametrize("upload", [0, 1, 2])
def test_group_union_equals_single_gpu_and_oracle(oracle, n_ranks, upload):
    """o2v_hip_group: materialless MAX (direct max-grid path) and textured BLEND with 2x supersampling."""
    from obj2voxel_amd import hip
    g = hip.DeviceGroup(_devices(n_ranks))
    single = hip.DeviceVoxelizer(0)
    try:
        assert g.comm_kind == ("rccl" if hip.device_count() >= n_ranks else "callbacks")
        v = meshes.uv_sphere(70)
        g.set_triangles(v, upload=upload)
        parts, cuts = g.voxelize(160, stage_times=True)
        assert cuts[0] == 0 and cuts[-1] == 160 and all(a < b for a, b in zip(cuts, cuts[1:]))
        for r, p in enumerate(parts):
            assert ((p[:, 2] >= cuts[r]) & (p[:, 2] < cuts[r + 1])).all()
        single.set_triangles(v)
        whole = meshes.sorted_voxels(single.voxelize(160))
        assert np.array_equal(meshes.sorted_voxels(np.concatenate(parts)), whole)
        assert np.array_equal(whole, meshes.sorted_voxels(oracle.voxelize(v, 160)))
        hits = [d.stats()["hits"] + d.stats()["skipped_jobs"] for d in g.ranks]
        # the plan balances predicted time - hits + 10 per leaf in occupancy-only mode, and the bands near the poles are many small
        # triangles - so the hits alone are only roughly equal
        assert max(hits) < 1.35 * sum(hits) / n_ranks, (cuts, hits)
        tm = g.ranks[0].timings()
        assert tm["plan_ms"] > 0 and tm["collective_ms"] > 0

        v, kw, tex = _textured_mesh()
        g.set_textures(tex)
        g.set_triangles(v, upload=upload, **kw)
        parts, cuts = g.voxelize(96, supersampling=2, strategy=1)
        want = meshes.sorted_voxels(oracle.voxelize(v, 96, supersampling=2, strategy=1, textures=tex, **kw))
        assert np.array_equal(meshes.sorted_voxels(np.concatenate(parts)), want)
        # a second job on the same group, user bounds and a unit transform: contexts and buffers are reused
        parts, cuts = g.voxelize(64, strategy=0, bounds=[-1.5, -1.5, -1.5, 1.5, 1.5, 1.5], unit_transform=[0, 0, 1, 0, -1, 0, 1, 0, 0])
        want = meshes.sorted_voxels(oracle.voxelize(v, 64, strategy=0, textures=tex, bounds=[-1.5, -1.5, -1.5, 1.5, 1.5, 1.5],
                                                    unit_transform=[0, 0, 1, 0, -1, 0, 1, 0, 0], **kw))
        assert np.array_equal(meshes.sorted_voxels(np.concatenate(parts)), want)
    finally:
        single.close()
        g.close()


def test_group_edge_cases(oracle):
    """More ranks than triangle blocks (ranks with an empty share), an empty mesh, a group of one."""
    from obj2voxel_amd import hip
    g = hip.DeviceGroup(_devices(4))
    try:
        v = meshes.unit_cube()                  # 12 triangles: one block, ranks 1..3 plan over nothing
        g.set_triangles(v)
        parts, cuts = g.voxelize(64)
        assert sum(len(p) for p in parts) == 23816    # reference test/main.cpp:120-126
        assert np.array_equal(meshes.sorted_voxels(np.concatenate(parts)), meshes.sorted_voxels(oracle.voxelize(v, 64)))
        g.set_triangles(np.zeros((0, 9), np.float32))
        counts, cuts = g.voxelize(64, read=False)
        assert counts == [0, 0, 0, 0]
    finally:
        g.close()
    g = hip.DeviceGroup([0])
    try:
        v = meshes.uv_sphere(9)
        g.set_triangles(v)
        parts, cuts = g.voxelize(48)
        assert cuts == [0, 48]
        assert np.array_equal(meshes.sorted_voxels(parts[0]), meshes.sorted_voxels(oracle.voxelize(v, 48)))
    finally:
        g.close()


def test_rccl_collectives_run_on_this_box(oracle, monkeypatch):
    """RCCL itself (librccl loaded with dlopen, ncclCommInitRank, all-reduce / all-gather on the context's stream): a
    communicator of one rank with the collectives forced on, which is all a single-GPU machine can host."""
    from obj2voxel_amd import hip
    monkeypatch.setenv("O2V_TEST_FORCE_COLLECTIVES", "1")
    comm = hip.Comm.rccl(hip.Comm.unique_id(), 0, 1, 0)
    d = hip.DeviceVoxelizer(0)
    try:
        assert comm.kind == "rccl"
        v = meshes.uv_sphere(20)
        d.set_triangles(v)
        got, counts, cuts = d.voxelize_sharded(comm, 100)
        assert counts == [len(got)] and cuts == [0, 100]
        assert d.timings()["collective_ms"] == 0        # timing a collective is a wait on the host: only on request
        got, counts, cuts = d.voxelize_sharded(comm, 100, stage_times=True)
        assert d.timings()["collective_ms"] > 0 and all(d.timings()["collective_parts_ms"][i] > 0 for i in (0, 2, 4))
        assert np.array_equal(meshes.sorted_voxels(got), meshes.sorted_voxels(oracle.voxelize(v, 100)))
    finally:
        d.close()
        comm.close()


def test_capi_voxelize_over_several_devices(oracle, monkeypatch):
    """obj2voxel_voxelize() with O2V_DEVICES naming several devices: same records as with one, fed to the sink rank by rank."""
    from obj2voxel_amd import capi
    a = capi.api()
    a.obj2voxel_set_log_level(capi.LOG_SILENT)
    a.o2v_release_cached_device_memory.restype = None
    a.o2v_release_cached_device_memory()
    v = meshes.uv_sphere(30)

    def run():
        inp, out = capi.TriangleInput(v), capi.CollectingOutput()
        inst = a.obj2voxel_alloc()
        a.obj2voxel_set_input_callback(inst, inp.callback, None)
        a.obj2voxel_set_output_callback(inst, out.callback, None)
        a.obj2voxel_set_resolution(inst, 120)
        assert a.obj2voxel_voxelize(inst) == 0
        a.obj2voxel_free(inst)
        return meshes.sorted_voxels(out.voxels())

    one = run()
    a.o2v_release_cached_device_memory()
    monkeypatch.setenv("O2V_DEVICES", ",".join(str(d) for d in _devices(3)))
    several = run()
    again = run()            # the cached group session is reused
    a.o2v_release_cached_device_memory()
    assert np.array_equal(one, several) and np.array_equal(one, again)
    assert np.array_equal(one, meshes.sorted_voxels(oracle.voxelize(v, 120)))


def _rank_main(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch  # noqa: F401  (first: the HIP runtime torch bundles must be the one that gets loaded)
    import torch.distributed as dist
    from obj2voxel_amd import hip
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    have = hip.device_count()
    d = hip.DeviceVoxelizer(rank % have)
    comm = hip.Comm.torch_distributed(dist)
    v, kw, tex = _textured_mesh()
    d.set_textures(tex)
    d.set_triangles(v, **kw)
    for strategy in (0, 1):
        vox, counts, cuts = d.voxelize_sharded(comm, 128, strategy=strategy)
        np.save(os.path.join(out_dir, f"vox{strategy}_{rank}.npy"), vox)
        np.save(os.path.join(out_dir, f"meta{strategy}_{rank}.npy"), np.array(counts + cuts, dtype=np.int64))
    dist.barrier()
    d.close()
    comm.close()
    dist.destroy_process_group()


def test_one_process_per_rank_gloo(tmp_path, oracle):
    """The layout bench.py --gpus N uses (one process per rank, torch.distributed), here with the gloo group supplying the
    collectives and the ranks sharing the GPU(s) of this box: o2v_hip_voxelize_sharded on device code."""
    import torch.multiprocessing as mp
    world = 2
    mp.spawn(_rank_main, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    v, kw, tex = _textured_mesh()
    for strategy in (0, 1):
        want = meshes.sorted_voxels(oracle.voxelize(v, 128, strategy=strategy, textures=tex, **kw))
        parts = [np.load(tmp_path / f"vox{strategy}_{r}.npy") for r in range(world)]
        meta = [np.load(tmp_path / f"meta{strategy}_{r}.npy") for r in range(world)]
        assert np.array_equal(meta[0], meta[1])                        # every rank gathered the same counts and cuts
        assert list(meta[0][:world]) == [len(p) for p in parts]
        cuts = list(meta[0][world:])
        for r, p in enumerate(parts):
            assert ((p[:, 2] >= cuts[r]) & (p[:, 2] < cuts[r + 1])).all()
        assert np.array_equal(meshes.sorted_voxels(np.concatenate(parts)), want)


def test_a_failing_rank_does_not_leave_the_others_in_a_collective(monkeypatch):
    """A rank that cannot prepare its sharded run (allocation failure, simulated by O2V_TEST_FAIL_RANK) reports it through the
    status all-reduce that precedes the planning collectives: every rank returns an error instead of waiting for it."""
    import threading
    from obj2voxel_amd import hip
    g = hip.DeviceGroup(_devices(3))
    try:
        g.set_triangles(meshes.uv_sphere(30))
        monkeypatch.setenv("O2V_TEST_FAIL_RANK", "1")
        result = {}

        def run():
            try:
                g.voxelize(64, read=False)
                result["ok"] = True
            except hip.DeviceError as e:
                result["err"] = str(e)
        t = threading.Thread(target=run)
        t.start()
        t.join(60)
        assert not t.is_alive(), "the group is stuck in a collective"
        assert "err" in result and "rank 1" in result["err"], result
        monkeypatch.delenv("O2V_TEST_FAIL_RANK")
        counts, cuts = g.voxelize(64, read=False)      # the group is usable afterwards
        assert sum(counts) > 0
    finally:
        g.close()


def test_an_exception_in_a_collective_callback_fails_the_call():
    """hip.Comm.torch_distributed wraps its callbacks: a Python exception inside one becomes a failed collective, not a
    silent success with un-reduced data (ctypes would print and swallow it and return 0)."""
    from obj2voxel_amd import hip

    class FakeDist:
        class ReduceOp:
            MIN, MAX, SUM = 0, 1, 2

        @staticmethod
        def get_rank():
            return 0

        @staticmethod
        def get_world_size():
            return 1

        @staticmethod
        def get_backend():
            return "gloo"

        @staticmethod
        def all_reduce(t, op=None):
            raise RuntimeError("simulated timeout")

        all_gather = broadcast = all_reduce
    import os
    os.environ["O2V_TEST_FORCE_COLLECTIVES"] = "1"
    comm = hip.Comm.torch_distributed(FakeDist)
    d = hip.DeviceVoxelizer(0)
    try:
        d.set_triangles(meshes.uv_sphere(10))
        with pytest.raises(hip.DeviceError, match="collective"):
            d.voxelize_sharded(comm, 32)
    finally:
        os.environ.pop("O2V_TEST_FORCE_COLLECTIVES", None)
        d.close()
        comm.close()


def test_block_list_gives_the_same_slabs(oracle, monkeypatch):
    """After a slab plan k_expand_roots visits only the listed blocks of 256 triangles that meet its slab (k_list_blocks):
    same records as without the list, for every slab; a plan made with other parameters must not be used."""
    from obj2voxel_amd import hip
    v = meshes.uv_sphere(90)         # 32 040 triangles = 126 blocks
    d = hip.DeviceVoxelizer(0)
    try:
        d.set_triangles(v)
        plain = [meshes.sorted_voxels(d.voxelize(200, zslab=z)) for z in ((0, 70), (70, 71), (71, 200))]
        monkeypatch.setenv("O2V_TEST_BLOCK_LIST", "1")
        cuts, bnd = d.plan_slabs(200, 3)
        listed = [meshes.sorted_voxels(d.voxelize(200, zslab=z)) for z in ((0, 70), (70, 71), (71, 200))]
        for a, b in zip(plain, listed):
            assert np.array_equal(a, b)
        assert d.stats()["leaves"] < len(v)           # the slab saw only its own triangles
        # another resolution: the extents belong to another transform, the list is not used (and the result is still right)
        other = meshes.sorted_voxels(d.voxelize(90, zslab=(10, 50)))
        want = meshes.sorted_voxels(oracle.voxelize(v, 90, zslab=(10, 50)))
        assert np.array_equal(other, want)
    finally:
        d.close()
