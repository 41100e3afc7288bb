#!/usr/bin/env python3
"""Predicts the balance of bench.py --gpus N on ONE GPU: runs the N-GPU job's planned z-slabs one after the other (same
mesh, same resolution, same cuts as o2v_hip_voxelize_sharded derives) and reports each slab's device time and voxel count.
The plan itself is timed as the full, unsharded passes (o2v_hip_plan_slabs): an upper bound, since the N ranks share those
passes (each streams 1/N of the triangles) and add two small all-reduces and two all-gathers instead.
predicted efficiency = (job voxels / (slowest slab + plan / N)) / (N * N=1 rate); the speed-up over the N = 1 bench job compares
two different jobs (the N = 1 line runs configs[2]), so the same-job strong scaling (sum of the slabs / slowest slab) is
printed beside it.  Neither is a measurement of an N-GPU run: xGMI and RCCL latencies are not in them.
usage: predict_scaling.py [N] [weak|config4]"""
import json
import sys
import time

sys.path.insert(0, '.')
from bench import workload_for
from obj2voxel_amd import hip, meshes

n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 8
kind = sys.argv[2] if len(sys.argv) > 2 else "weak"
steps = 5
dv = hip.DeviceVoxelizer(0)
_, res1, nv1 = workload_for(1)
dv.set_triangles(meshes.uv_sphere(nv1))
for _ in range(3):
    dv.voxelize(res1, read=False)
t0 = time.perf_counter()
for _ in range(steps):
    v1 = dv.voxelize(res1, read=False)
t1 = (time.perf_counter() - t0) / steps * 1e3
name, res, nv = workload_for(n, kind)
verts = meshes.uv_sphere(nv)
dv.set_triangles(verts)
for _ in range(2):
    cuts, bnd = dv.plan_slabs(res, n)
t0 = time.perf_counter()
for _ in range(steps):
    cuts, bnd = dv.plan_slabs(res, n)
plan_ms = (time.perf_counter() - t0) / steps * 1e3
rows = []
for r in range(n):
    for _ in range(2):
        dv.voxelize(res, zslab=(cuts[r], cuts[r + 1]), bounds=bnd, read=False)
    walls = []
    for i in range(2 * steps - 1):
        t0 = time.perf_counter()
        cnt = dv.voxelize(res, zslab=(cuts[r], cuts[r + 1]), bounds=bnd, read=False)
        walls.append((time.perf_counter() - t0) * 1e3)
    wall = sorted(walls)[len(walls) // 2]   # median: a single host hiccup must not decide the slowest slab
    dv.voxelize(res, zslab=(cuts[r], cuts[r + 1]), bounds=bnd, read=False, stage_times=True)   # (the stage times: one more pass)
    st, t = dv.stats(), dv.timings()
    rows.append({"rank": r, "z": [cuts[r], cuts[r + 1]], "voxels": cnt, "leaves": st["leaves"], "hits": st["hits"],
                 "ms": round(wall, 3), "stages": {k: round(v, 3) for k, v in t.items() if k.endswith("_ms") and isinstance(v, float)}})
    print(json.dumps(rows[-1]), flush=True)
worst = max(r["ms"] for r in rows) + plan_ms / n
total = sum(r["voxels"] for r in rows)
print(json.dumps({"n": n, "workload": name, "resolution": res, "triangles": len(verts), "total_voxels": total,
                  "n1_ms": round(t1, 3), "n1_mvoxels_per_s": round(v1 / t1 / 1e3, 1), "full_plan_ms": round(plan_ms, 3),
                  "max_slab_ms": max(r["ms"] for r in rows), "mean_slab_ms": round(sum(r["ms"] for r in rows) / n, 3),
                  "predicted_step_ms": round(worst, 3), "predicted_mvoxels_per_s": round(total / worst / 1e3, 1),
                  "predicted_speedup_over_n1": round((total / worst) / (v1 / t1), 2),
                  "predicted_weak_scaling_efficiency": round((total / worst) / (n * v1 / t1), 3),
                  # the same job on one GPU = its slabs one after the other (the whole grid may not even fit one GPU)
                  "same_job_one_gpu_ms": round(sum(r["ms"] for r in rows) + plan_ms, 3),
                  "predicted_same_job_strong_scaling": round((sum(r["ms"] for r in rows) + plan_ms) / worst, 2)}))
