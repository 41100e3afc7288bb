#!/bin/bash
# Developer tool: time the bench with ablation builds of the library (obj2voxel_amd/libvar*.so).
cd "$(dirname "$0")/.."
cp obj2voxel_amd/libobj2voxel_amd.so /tmp/lib_orig.so
for v in "$@"; do
  cp obj2voxel_amd/libvar$v.so obj2voxel_amd/libobj2voxel_amd.so
  echo "== variant $v"
  timeout -k 5 100 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['pipeline']['stages_ms'], d['config']['voxels'])"
done
cp /tmp/lib_orig.so obj2voxel_amd/libobj2voxel_amd.so
