#!/bin/bash
# Developer tool: per-kernel average times of one tools/bench_configs.py case (substring of its name).
cd "$(dirname "$0")/.."
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rm -rf /tmp/prof_c
timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c -o x -- python tools/bench_configs.py "$1" > /tmp/prof_c.log 2>&1
grep "^{" /tmp/prof_c.log | cut -c1-300
python - <<PY
import csv,re
for r in csv.DictReader(open("/tmp/prof_c/x_kernel_stats.csv")):
    n=r["Name"]; m=re.search(r"(k_[a-z_]+(<[^>]*>)?)", n)
    if m and float(r["AverageNs"])>20000: print("   %-34s calls=%4s avg_us=%9.1f" % (m.group(1), r["Calls"], float(r["AverageNs"])/1e3))
PY
