#!/bin/bash
# Developer tool: dynamic VALU instruction count / lane utilisation of k_voxelize on the bench workload.
cd "$(dirname "$0")/.."
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rm -rf /tmp/pmc_k2
timeout -k 5 150 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY --output-format csv -d /tmp/pmc_k2 -o p -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python - <<PY
import csv,re,collections
acc=collections.defaultdict(list)
for r in csv.DictReader(open("/tmp/pmc_k2/p_counter_collection.csv")):
    if "k_voxelize" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
d={k: sum(v)/len(v) for k,v in acc.items()}
print({k: round(v/1e6,1) for k,v in d.items()}, "(millions)")
print("active lanes/VALU instr: %.1f   VALU active frac per wave: %.3f" % (d["SQ_THREAD_CYCLES_VALU"]/d["SQ_INSTS_VALU"], d["SQ_ACTIVE_INST_VALU"]/d["SQ_WAVE_CYCLES"]))
PY
