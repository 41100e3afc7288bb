"""The N > 1 path without a GPU: two processes (gloo) and in-process threads.

What runs here is the host side of the sharded voxelization (include/o2v_hip.h, multi-GPU section): the slab ranges tile
the grid; the collectives the ranks use - torch.distributed gloo callbacks between processes, the shared-memory exchange
between the threads of an in-process group - behave as specified (the library's own self-test drives them with known
patterns); partial work histograms summed over the ranks give every rank the same cuts as the whole histogram.  The
device side of the same path (two ranks sharing one GPU, results against the oracle) is tests/test_gpu_multi.py.
The oracle only appears as the reference for what a z-slab of the result is.
"""
import ctypes as C
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _cuts(hist, bin_layers, res, n):
    from obj2voxel_amd import hip
    L = hip._bind()
    L.o2v_hip_cuts_from_histogram.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
    hist = np.ascontiguousarray(hist, dtype=np.uint64)
    out = np.zeros(n + 1, dtype=np.uint32)
    L.o2v_hip_cuts_from_histogram(hist.ctypes.data, len(hist), bin_layers, res, n, out.ctypes.data)
    return [int(z) for z in out]


def _worker(rank, world, port, res, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    from obj2voxel_amd import hip, meshes, slab
    from oracle import oracle
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    # 1. the gloo-backed collectives, driven by the library's self-test (host memory only)
    comm = hip.Comm.torch_distributed(dist)
    L = hip._bind()
    L.o2v_hip_comm_callbacks_selftest.argtypes = [C.c_void_p, C.c_int, C.c_int]
    selftest = L.o2v_hip_comm_callbacks_selftest(C.byref(comm._keep), rank, world)
    # 2. sharded planning: every rank histograms its share of the triangles, the sum gives everyone the same cuts
    rng = np.random.default_rng(11)            # same seed on every rank: the "whole" histogram
    whole = rng.integers(0, 1 << 30, size=res, dtype=np.uint64)
    share = whole // world + (np.arange(res) % world == rank) * (whole % world)   # shares that add up to `whole`
    t = torch.from_numpy(share.astype(np.int64))
    dist.all_reduce(t)
    cuts = _cuts(t.numpy().astype(np.uint64), 1, res, world)
    # 3. what a slab is: the oracle's voxels with z in the rank's range
    v = meshes.uv_sphere(12)
    T = len(v)
    z0, z1 = slab.slab_range(rank, world, res)
    vox = oracle.voxelize(v, res, types=np.full(T, 2, np.uint32), colors=meshes.triangle_colors(T), strategy=1,
                          zslab=(z0, z1))
    total, tmax = slab.reduce_job(dist, len(vox), 0.5 + rank)
    np.save(os.path.join(out_dir, f"slab{rank}.npy"), vox)
    np.save(os.path.join(out_dir, f"meta{rank}.npy"), np.array([selftest, total, tmax] + cuts, dtype=np.float64))
    dist.barrier()
    comm.close()
    dist.destroy_process_group()


def test_slab_ranges_tile_the_grid():
    from obj2voxel_amd import slab
    for world in (1, 2, 3, 8):
        for res in (8, 100, 1024, 2896):
            edges = [slab.slab_range(r, world, res) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == res
            assert all(a[1] == b[0] for a, b in zip(edges, edges[1:]))
    with pytest.raises(ValueError):
        slab.slab_range(2, 2, 64)


def test_cuts_from_histogram_balance_and_edge_cases():
    res = 512
    flat = np.full(res, 1000, np.uint64)
    assert _cuts(flat, 1, res, 4) == [0, 128, 256, 384, 512]
    assert _cuts(np.zeros(res, np.uint64), 1, res, 4) == [0, 128, 256, 384, 512]      # no work: equal heights
    ramp = np.arange(res, dtype=np.uint64) + 1                                          # work grows with z
    cuts = _cuts(ramp, 1, res, 8)
    assert cuts[0] == 0 and cuts[-1] == res and all(a < b for a, b in zip(cuts, cuts[1:]))
    work = [int(ramp[a:b].sum()) for a, b in zip(cuts, cuts[1:])]
    assert max(work) < 1.03 * sum(work) / 8
    spike = np.zeros(res, np.uint64)
    spike[300] = 10 ** 12                                                                # all the work in one layer
    cuts = _cuts(spike, 1, res, 4)
    assert cuts[0] == 0 and cuts[-1] == res and all(a < b for a, b in zip(cuts, cuts[1:]))
    bins = np.full(256, 5, np.uint64)                                                    # bins of 4 layers
    assert _cuts(bins, 4, 1024, 2) == [0, 512, 1024]
    assert _cuts(flat[:16], 1, 16, 16) == list(range(17))                                # one layer per slab


@pytest.mark.parametrize("n_threads", [2, 3, 8])
def test_shared_memory_exchange_between_threads(n_threads):
    """The collectives of an in-process group when RCCL cannot be used (threads, host memory), no GPU involved."""
    from obj2voxel_amd import hip
    L = hip._bind()
    L.o2v_hip_group_exchange_selftest.argtypes = [C.c_uint32]
    assert L.o2v_hip_group_exchange_selftest(n_threads) == 0


def test_two_rank_gloo_collectives_plan_and_slabs(tmp_path, oracle):
    import torch.multiprocessing as mp
    from obj2voxel_amd import meshes
    res, world = 96, 2
    mp.spawn(_worker, args=(world, _free_port(), res, str(tmp_path)), nprocs=world, join=True)
    v = meshes.uv_sphere(12)
    T = len(v)
    full = meshes.sorted_voxels(oracle.voxelize(v, res, types=np.full(T, 2, np.uint32),
                                                colors=meshes.triangle_colors(T), strategy=1))
    parts = [np.load(tmp_path / f"slab{r}.npy") for r in range(world)]
    meta = [np.load(tmp_path / f"meta{r}.npy") for r in range(world)]
    for m in meta:
        assert m[0] == 0, f"collective self-test failed at step {int(m[0])}"
        assert int(m[1]) == len(full) == sum(len(p) for p in parts)
        assert m[2] == 1.5  # max over ranks
    rng = np.random.default_rng(11)
    whole = rng.integers(0, 1 << 30, size=res, dtype=np.uint64)
    assert list(meta[0][3:]) == list(meta[1][3:]) == _cuts(whole, 1, res, world)       # same cuts on every rank
    assert np.array_equal(meshes.sorted_voxels(np.concatenate(parts)), full)
    for r, p in enumerate(parts):
        assert ((p[:, 2] >= r * 48) & (p[:, 2] < (r + 1) * 48)).all()
