#!/bin/bash
# Developer tool: per-kernel times of the bench with ablation builds of the library (obj2voxel_amd/libvar*.so).
cd "$(dirname "$0")/.."
cp obj2voxel_amd/libobj2voxel_amd.so /tmp/lib_orig.so
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for v in "$@"; do
  cp obj2voxel_amd/libvar$v.so obj2voxel_amd/libobj2voxel_amd.so
  echo "== variant $v"
  rm -rf /tmp/abl_$v
  timeout -k 5 100 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abl_$v -o x -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
  python - <<PY
import csv,re
for r in csv.DictReader(open("/tmp/abl_$v/x_kernel_stats.csv")):
    n=r["Name"]; m=re.search(r"(k_[a-z_]+(<[^>]*>)?)", n)
    if m and float(r["AverageNs"])>20000: print("   %-34s avg_us=%8.1f" % (m.group(1), float(r["AverageNs"])/1e3))
PY
done
cp /tmp/lib_orig.so obj2voxel_amd/libobj2voxel_amd.so
