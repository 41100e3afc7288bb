#!/bin/bash
# Runs on the GPU box: every rocprofv3 pass the committed summaries under profiles/<round>/ are made from
# (tools/collect_profiles.py turns gpurun_out/ into profiles/).  Counter passes use --kernel-trace only.
cd "$(dirname "$0")/.."
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
OUT=gpurun_out
mkdir -p $OUT
B="python bench.py --no-cpu-baseline"
timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_final -o f -- $B --steps 20 --warmup 3 > $OUT/prof_final.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 5 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmcf_$c -o p -- $B --steps 3 --warmup 1 > $OUT/pmcf_$c.log 2>&1
done
timeout -k 5 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmcf_sq -o p -- $B --steps 3 --warmup 1 > $OUT/pmcf_sq.log 2>&1
# the bench line itself (with the CPU baseline), unprofiled
timeout -k 5 600 python bench.py > $OUT/bench_final.json 2> $OUT/bench_final.err
# predicted multi-GPU balance (one GPU runs the 8 slabs of the 8-GPU job in turn)
timeout -k 5 300 python tools/predict_scaling.py 8 > $OUT/predict_scaling_8.jsonl 2>&1
find $OUT -name "*_kernel_stats.csv" -o -name "*_counter_collection.csv" | head
tail -c 600 $OUT/bench_final.json
