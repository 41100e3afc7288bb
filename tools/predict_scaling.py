#!/usr/bin/env python3
"""Predicts the weak-scaling balance of bench.py --gpus N on ONE GPU: runs the N-GPU job's z-slabs one after the other
(same mesh, same resolution, same slab plan as bench.py) and reports each slab's wall time per step (plan + voxelize,
as bench.py times it) and voxel count.  predicted efficiency = (job voxels / slowest slab) / (N * N=1 rate).
usage: predict_scaling.py [N] [--equal]   (--equal: equal-height slabs instead of o2v_hip_plan_slabs)"""
import json
import sys
import time

sys.path.insert(0, '.')
from bench import workload_for
from obj2voxel_amd import hip, meshes, slab

n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 8
equal = "--equal" in sys.argv
steps = 5
dv = hip.DeviceVoxelizer(0)
res1, nv1 = workload_for(1)
dv.set_triangles(meshes.uv_sphere(nv1))
for _ in range(3):
    dv.voxelize(res1, read=False)
t0 = time.perf_counter()
for _ in range(steps):
    v1 = dv.voxelize(res1, read=False)
t1 = (time.perf_counter() - t0) / steps * 1e3
res, nv = workload_for(n)
verts = meshes.uv_sphere(nv)
dv.set_triangles(verts)
rows = []
def step(r):
    if equal:
        z0, z1 = slab.slab_range(r, n, res)
        return dv.voxelize(res, zslab=(z0, z1), read=False), (z0, z1)
    cuts, bnd = dv.plan_slabs(res, n)
    return dv.voxelize(res, zslab=(cuts[r], cuts[r + 1]), bounds=bnd, read=False), (cuts[r], cuts[r + 1])


for r in range(n):
    for _ in range(2):
        step(r)
    t0 = time.perf_counter()
    for i in range(steps):
        cnt, (z0, z1) = step(r)
    wall = (time.perf_counter() - t0) / steps * 1e3
    st, t = dv.stats(), dv.timings()
    rows.append({"rank": r, "z": [z0, z1], "voxels": cnt, "leaves": st["leaves"], "hits": st["hits"],
                 "ms": round(wall, 3), "stages": {k: round(v, 3) for k, v in t.items() if k.endswith("_ms")}})
    print(json.dumps(rows[-1]), flush=True)
worst = max(r["ms"] for r in rows)
print(json.dumps({"n": n, "resolution": res, "triangles": len(verts), "total_voxels": sum(r["voxels"] for r in rows),
                  "n1_ms": round(t1, 3), "max_slab_ms": worst, "mean_slab_ms": round(sum(r["ms"] for r in rows) / n, 3),
                  "predicted_weak_scaling_efficiency": round((sum(r["voxels"] for r in rows) / worst) / (n * v1 / t1), 3)}))
