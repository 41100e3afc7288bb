#!/bin/bash
# Developer tool: hardware counters of the pipeline's kernels on one workload.
# usage: tools/pmc.sh WORKLOAD OUT_PREFIX "COUNTER ..." ["COUNTER ..." ...]     (O2V_LIB selects the library,
#        O2V_PMC_KERNELS a regular expression of the kernels to report: default k_voxelize and k_scatter)
cd "$(dirname "$0")/.."
ROOT=$PWD
W=$1; OUT=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
i=0
for set in "$@"; do
  i=$((i+1))
  rm -rf /tmp/pmc_x
  timeout -k 5 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmc_x -o p -- python tools/run_workload.py $W --steps 2 --warmup 1 > /tmp/pmc_x.log 2>&1
  python - "$OUT" <<'PY'
import csv, collections, re, sys, glob
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/pmc_x/**/p_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_[a-z_0-9]+(<[^>]*>)?)", r["Kernel_Name"])
        if m: acc[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(sys.argv[1] + ".txt", "a") as out:
    for k, d in acc.items():
        import os
        if not re.search(os.environ.get("O2V_PMC_KERNELS", "^k_voxelize|^k_scatter"), k): continue
        print(k, {c: round(sum(v) / len(v) / 1e6, 2) for c, v in d.items()}, "(millions per launch)", file=out)
PY
done
