"""Developer tool: dissect fuzz cases (tests/test_gpu_fuzz.py) that differ between device and oracle."""
import sys
sys.path.insert(0, '.')
import numpy as np
from obj2voxel_amd import hip, meshes
from oracle import oracle
from tests.test_gpu_fuzz import _case

dv = hip.DeviceVoxelizer(0)
for seed in [int(a) for a in sys.argv[1:]]:
    v, res, kw, mat, textures = _case(seed)
    dv.set_textures(textures)
    dv.set_triangles(v, **mat)
    g = meshes.sorted_voxels(dv.voxelize(res, **kw))
    w = meshes.sorted_voxels(oracle.voxelize(v, res, textures=textures, **mat, **kw))
    print("seed", seed, "T", len(v), "res", res, kw, "dev", len(g), "oracle", len(w))
    gs = {tuple(r[:3]) for r in g.tolist()}
    ws = {tuple(r[:3]) for r in w.tolist()}
    print("  only dev", len(gs - ws), sorted(gs - ws)[:5], " only oracle", len(ws - gs), sorted(ws - gs)[:5])
    # try removing options one by one
    for drop in ("zslab", "bounds", "unit_transform", "supersampling"):
        if drop in kw:
            k2 = {k: x for k, x in kw.items() if k != drop}
            g2 = dv.voxelize(res, **k2)
            w2 = oracle.voxelize(v, res, textures=textures, **mat, **k2)
            print("   without", drop, "dev", len(g2), "oracle", len(w2))
