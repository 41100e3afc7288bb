#!/usr/bin/env python3
"""BASELINE.json configs[4] on one GPU: the 50 M-triangle sphere (nv = 3536) at 4096^3 split into 8 planned z-slabs;
runs the given ranks' slabs one after the other (what each of the 8 GPUs would do) and prints timings and counters.
usage: config5_slab.py [rank ...]   (default: 0 3)"""
import json
import sys
import time

sys.path.insert(0, '.')
from obj2voxel_amd import hip, meshes

ranks = [int(a) for a in sys.argv[1:]] or [0, 3]
t0 = time.perf_counter()
verts = meshes.uv_sphere(3536)
print(f"mesh: {len(verts)} triangles, generated in {time.perf_counter() - t0:.1f} s", flush=True)
dv = hip.DeviceVoxelizer(0)
t0 = time.perf_counter()
dv.set_triangles(verts)
print(f"upload: {time.perf_counter() - t0:.2f} s", flush=True)
res, n = 4096, 8
cuts, bnd = dv.plan_slabs(res, n)
print("cuts", cuts, flush=True)
for r in ranks:
    for i in range(3):
        t0 = time.perf_counter()
        cuts, bnd = dv.plan_slabs(res, n)
        cnt = dv.voxelize(res, zslab=(cuts[r], cuts[r + 1]), bounds=bnd, read=False)
        wall = (time.perf_counter() - t0) * 1e3
    st, tm = dv.stats(), dv.timings()
    print(json.dumps({"rank": r, "z": [cuts[r], cuts[r + 1]], "voxels": cnt, "wall_ms": round(wall, 3),
                      "stages": {k: round(v, 3) for k, v in tm.items()}, "leaves": st["leaves"], "hits": st["hits"],
                      "direct_hits": st["direct_hits"], "grid_GB": round(st["grid_bytes"] / 1e9, 1)}), flush=True)
