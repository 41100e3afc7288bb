#!/bin/bash
# Developer tool: per-kernel average times of the bench workload (rocprofv3 --kernel-trace --stats).
cd "$(dirname "$0")/.."
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rm -rf /tmp/prof_x
timeout -k 5 150 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_x -o x -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" > /tmp/prof_x.log 2>&1
grep "^{" /tmp/prof_x.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['pipeline']['stages_ms'])"
python - <<PY
import csv,re
for r in csv.DictReader(open("/tmp/prof_x/x_kernel_stats.csv")):
    n=r["Name"]; m=re.search(r"(k_[a-z_]+(<[^>]*>)?)", n)
    if m and float(r["AverageNs"])>10000: print("   %-34s avg_us=%8.1f" % (m.group(1), float(r["AverageNs"])/1e3))
PY
