"""The N > 1 path on CPU: two processes (gloo), each owning a z-slab. The per-slab worker is the CPU oracle
(there is no GPU here); what is under test is the sharding logic bench.py uses (obj2voxel_amd.slab): slab ranges
tile the grid, the summed count equals the single-process run, and the union of the slabs is bit-identical."""
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


def _worker(rank, world, port, res, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from obj2voxel_amd import meshes, slab
    from oracle import oracle
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    v = meshes.uv_sphere(12)
    T = len(v)
    z0, z1 = slab.slab_range(rank, world, res)
    vox = oracle.voxelize(v, res, types=np.full(T, 2, np.uint32), colors=meshes.triangle_colors(T), strategy=1,
                          zslab=(z0, z1))
    total, t = slab.reduce_job(dist, len(vox), 0.5 + rank)
    np.save(os.path.join(out_dir, f"slab{rank}.npy"), vox)
    if rank == 0:
        np.save(os.path.join(out_dir, "total.npy"), np.array([total, t]))
    dist.barrier()
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


def test_two_rank_slabs_equal_single_run(tmp_path, oracle):
    import torch.multiprocessing as mp
    from obj2voxel_amd import meshes
    res, world = 96, 2
    mp.spawn(_worker, args=(world, _free_port(), res, str(tmp_path)), nprocs=world, join=True)
    v = meshes.uv_sphere(12)
    T = len(v)
    full = meshes.sorted_voxels(oracle.voxelize(v, res, types=np.full(T, 2, np.uint32),
                                                colors=meshes.triangle_colors(T), strategy=1))
    parts = [np.load(tmp_path / f"slab{r}.npy") for r in range(world)]
    total, t = np.load(tmp_path / "total.npy")
    assert int(total) == len(full) == sum(len(p) for p in parts)
    assert t == 1.5  # max over ranks
    assert np.array_equal(meshes.sorted_voxels(np.concatenate(parts)), full)
    for r, p in enumerate(parts):
        assert ((p[:, 2] >= r * 48) & (p[:, 2] < (r + 1) * 48)).all()
