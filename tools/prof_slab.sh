#!/bin/bash
# Developer tool: per-kernel average times of ONE planned slab of the N-GPU weak-scaling job (rank R of N), run on one GPU.
# usage: tools/prof_slab.sh N R
cd "$(dirname "$0")/.."
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rm -rf /tmp/prof_s
cat > /tmp/prof_slab.py <<PY
import sys, time
sys.path.insert(0, '.')
from bench import workload_for
from obj2voxel_amd import hip, meshes
n, r = $1, $2
res, nv = workload_for(n)
dv = hip.DeviceVoxelizer(0)
dv.set_triangles(meshes.uv_sphere(nv))
for i in range(12):
    if i == 2: t0 = time.perf_counter()
    cuts, bnd = dv.plan_slabs(res, n)
    tp = time.perf_counter()
    dv.voxelize(res, zslab=(cuts[r], cuts[r + 1]), bounds=bnd, read=False)
    if i == 11: print("last plan wall ms", (tp - tl) * 1e3) if False else None
print("wall ms/step", (time.perf_counter() - t0) / 10 * 1e3, cuts, dv.timings())
t0 = time.perf_counter()
for i in range(10):
    dv.plan_slabs(res, n)
print("plan wall ms", (time.perf_counter() - t0) / 10 * 1e3)
PY
timeout -k 5 150 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_s -o x -- python /tmp/prof_slab.py > /tmp/prof_s.log 2>&1
grep "wall ms" /tmp/prof_s.log
python - <<PY
import csv,re
for r in csv.DictReader(open("/tmp/prof_s/x_kernel_stats.csv")):
    n=r["Name"]; m=re.search(r"(k_[a-z_]+(<[^>]*>)?)", n)
    if m and float(r["AverageNs"])>5000: print("   %-34s calls=%4s avg_us=%8.1f" % (m.group(1), r["Calls"], float(r["AverageNs"])/1e3))
PY
