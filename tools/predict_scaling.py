#!/usr/bin/env python3
"""Predicts the weak-scaling balance of bench.py --gpus N on ONE GPU: runs the N-GPU job's z-slabs one after the other
(same mesh, same resolution, same slab ranges as bench.py) and reports each slab's device time and voxel count.
predicted efficiency vs. N=1 = t(N=1 job) / max_slab_time."""
import json
import sys
import time

sys.path.insert(0, '.')
from bench import workload_for
from obj2voxel_amd import hip, meshes, slab

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = 5
dv = hip.DeviceVoxelizer(0)
res1, nv1 = workload_for(1)
dv.set_triangles(meshes.uv_sphere(nv1))
for _ in range(3):
    dv.voxelize(res1, read=False)
t1 = dv.timings()["total_ms"]
res, nv = workload_for(n)
verts = meshes.uv_sphere(nv)
dv.set_triangles(verts)
rows = []
for r in range(n):
    z0, z1 = slab.slab_range(r, n, res)
    for _ in range(2):
        dv.voxelize(res, zslab=(z0, z1), read=False)
    tm = [0.0] * steps
    for i in range(steps):
        cnt = dv.voxelize(res, zslab=(z0, z1), read=False)
        tm[i] = dv.timings()["total_ms"]
    st, t = dv.stats(), dv.timings()
    rows.append({"rank": r, "z": [z0, z1], "voxels": cnt, "leaves": st["leaves"], "hits": st["hits"],
                 "ms": round(sum(tm) / steps, 3), "stages": {k: round(v, 3) for k, v in t.items() if k.endswith("_ms")}})
    print(json.dumps(rows[-1]), flush=True)
worst = max(r["ms"] for r in rows)
print(json.dumps({"n": n, "resolution": res, "triangles": len(verts), "total_voxels": sum(r["voxels"] for r in rows),
                  "n1_ms": round(t1, 3), "max_slab_ms": worst, "mean_slab_ms": round(sum(r["ms"] for r in rows) / n, 3),
                  "predicted_weak_scaling_efficiency": round((sum(r["voxels"] for r in rows) / worst) / (n * 4936186 / t1), 3)}))
