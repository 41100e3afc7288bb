#!/bin/bash
# Developer tool: wait / issue breakdown of k_voxelize on the bench workload (two rocprofv3 --pmc passes).
cd "$(dirname "$0")/.."
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC" \
           "SQ_INSTS_BRANCH SQ_IFETCH SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_TRANS_F32 SQ_BUSY_CYCLES SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_LDS_IDX_ACTIVE"; do
  rm -rf /tmp/pmc_k2b
  timeout -k 5 150 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmc_k2b -o p -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  python - <<PY
import csv,collections
acc=collections.defaultdict(list)
for r in csv.DictReader(open("/tmp/pmc_k2b/p_counter_collection.csv")):
    if "k_voxelize" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print({k: round(sum(v)/len(v)/1e6,1) for k,v in acc.items()}, "(millions)")
PY
done
