#!/bin/bash
# Developer tool: A/B the current library against obj2voxel_amd/libvar<NAME>.so on bench.py and bench_configs.py.
cd "$(dirname "$0")/.."
run() {
  timeout -k 5 100 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  bench', d['value'], d['pipeline']['stages_ms'])"
  timeout -k 5 200 python tools/bench_configs.py 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('  %-70s %8.3f ms  K2 %.3f' % (d['case'][:70], d['ms'], d['stages_ms']['voxelize_ms']))"
}
echo "== current"; run
cp obj2voxel_amd/libobj2voxel_amd.so /tmp/lib_orig.so
for v in "$@"; do
  cp obj2voxel_amd/libvar$v.so obj2voxel_amd/libobj2voxel_amd.so
  echo "== variant $v"; run
done
cp /tmp/lib_orig.so obj2voxel_amd/libobj2voxel_amd.so
