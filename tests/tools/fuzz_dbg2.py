"""Developer tool: dissect planar-stress fuzz cases."""
import sys
sys.path.insert(0, '.')
import numpy as np
from obj2voxel_amd import hip, meshes
from oracle import oracle
from tests.test_gpu_fuzz import _planar_case

dv = hip.DeviceVoxelizer(0)
for seed in [int(a) for a in sys.argv[1:]]:
    v, res, kw, mat = _planar_case(seed)
    dv.set_triangles(v, **mat)
    g = meshes.sorted_voxels(dv.voxelize(res, **kw))
    w = meshes.sorted_voxels(oracle.voxelize(v, res, **mat, **kw))
    print("seed", seed, "T", len(v), "res", res, "dev", len(g), "oracle", len(w))
    print("  xform dev", dv.transform(), "oracle", oracle.mesh_transform(kw["bounds"], res))
    gs = {tuple(r[:3]) for r in g.tolist()}
    ws = {tuple(r[:3]) for r in w.tolist()}
    print("  only dev", len(gs - ws), sorted(gs - ws)[:6], " only oracle", len(ws - gs), sorted(ws - gs)[:6])
    if len(g) == len(w) and gs == ws:
        bad = np.flatnonzero(g[:, 3] != w[:, 3])
        print("  colour mismatches", len(bad), [(g[b].tolist(), hex(w[b, 3])) for b in bad[:4]])
    # single triangles
    for t in range(len(v)):
        dv.set_triangles(v[t:t + 1], types=mat["types"][t:t + 1], colors=mat["colors"][t:t + 1])
        g1 = dv.voxelize(res, **kw)
        w1 = oracle.voxelize(v[t:t + 1], res, types=mat["types"][t:t + 1], colors=mat["colors"][t:t + 1], **kw)
        if len(g1) != len(w1):
            print("   tri", t, "dev", len(g1), "oracle", len(w1), v[t].tolist())
            break
