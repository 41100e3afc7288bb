#!/bin/bash
# usage: tools/profile_all.sh [workload ...]   (default: every profiled workload)
# Runs on the GPU box: every rocprofv3 pass the committed summaries under profiles/<round>/ are made from
# (tools/collect_profiles.py turns gpurun_out/prof/ into profiles/).  Counter passes use --kernel-trace only, one counter
# family per pass.  Workloads: the bench command itself (BASELINE configs[2], direct MAX route) and the general route
# (pool -> counting sort -> ordered replay): the bench mesh with BLEND, configs[1], configs[3] (tools/run_workload.py).
cd "$(dirname "$0")/.."
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
OUT=gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
SQ1="SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY"
SQ2="SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_BUSY_CYCLES SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS"
# front end and pipes (instruction fetches, scalar unit cycles, the second VALU pipe, transcendentals, SE cycles)
SQ3="SQ_IFETCH SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU2 SQ_INSTS_VALU_TRANS_F32 SQ_INST_LEVEL_VMEM"
WL="${@:-config2 config2_colored_max config2_blend config2_textured_max scan_colored_max config1 config3}"
for W in $WL; do
  if [ $W = config2 ]; then CMD="python bench.py --no-cpu-baseline --no-capi --no-routes"; S1="--steps 20 --warmup 3"; S2="--steps 3 --warmup 1"
  else CMD="python tools/run_workload.py $W"; S1="--steps 10 --warmup 2"; S2="--steps 3 --warmup 1"; fi
  timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${W}_stats -o s -- $CMD $S1 > $OUT/${W}_stats.log 2>&1
  grep "^{" $OUT/${W}_stats.log | tail -1 > $OUT/${W}_line.json
  # (bench.py's stdout line is the compact object: the device statistics collect_profiles.py wants are in its details file)
  [ $W = config2 ] && [ -s bench_details.json ] && cp bench_details.json $OUT/${W}_line.json
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 5 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/${W}_$c -o p -- $CMD $S2 > $OUT/${W}_$c.log 2>&1
  done
  timeout -k 5 300 rocprofv3 --kernel-trace --pmc $SQ1 --output-format csv -d $OUT/${W}_sq1 -o p -- $CMD $S2 > $OUT/${W}_sq1.log 2>&1
  timeout -k 5 300 rocprofv3 --kernel-trace --pmc $SQ2 --output-format csv -d $OUT/${W}_sq2 -o p -- $CMD $S2 > $OUT/${W}_sq2.log 2>&1
  if [ $W = config2 ] || [ $W = config3 ] || [ $W = config2_blend ]; then
    timeout -k 5 300 rocprofv3 --kernel-trace --pmc $SQ3 --output-format csv -d $OUT/${W}_sq3 -o p -- $CMD $S2 > $OUT/${W}_sq3.log 2>&1
  fi
done
python -c "from obj2voxel_amd import hip; print(hip.build_id())" > $OUT/build_id.txt
# issue costs of the instructions the clip loop is made of (make -C tools/ubench here, before the gpurun call)
[ -x tools/ubench/_build/valu_rates ] && timeout -k 5 60 tools/ubench/_build/valu_rates > $OUT/valu_rates.json 2>/dev/null
# the summaries are made here, so that the bench line below reads the counters of THIS build (profiles/current.json), and
# travel back under gpurun_out/ (copy gpurun_out/prof/profiles/* into profiles/ afterwards)
ROUND=${O2V_ROUND:-r06}
mkdir -p profiles/$ROUND
# the clip loop's instruction histogram priced with the measured issue costs (the mix-weighted ceiling of bench.py's roofline)
RATES=profiles/r05/valu_rates.json; [ -s $OUT/valu_rates.json ] && cp $OUT/valu_rates.json profiles/$ROUND/valu_rates.json && RATES=profiles/$ROUND/valu_rates.json
python tools/isa_hist.py --rates $RATES --json profiles/$ROUND/isa_hist.json > $OUT/isa_hist.txt 2>&1 && cp profiles/$ROUND/isa_hist.json profiles/current_isa.json
python tools/collect_profiles.py $ROUND > $OUT/collect.log 2>&1
# the bench line itself (with the routes, the CPU baseline and the C API wall time), unprofiled
timeout -k 5 900 python bench.py > $OUT/bench_line.json 2> $OUT/bench_line.err
cp $OUT/bench_line.json profiles/$ROUND/bench_line.json
# (the stdout line is the compact object the driver parses; everything else bench.py gathered is in its details file)
cp bench_details.json profiles/$ROUND/bench_details.json 2>/dev/null
# predicted multi-GPU balance (one GPU runs the planned slabs of the N = 8 jobs one after the other)
timeout -k 5 200 python tools/predict_scaling.py 8 weak > $OUT/predict_scaling_8_weak.jsonl 2>&1
timeout -k 5 500 python tools/predict_scaling.py 8 config4 > $OUT/predict_scaling_8_config4.jsonl 2>&1
cp $OUT/predict_scaling_8_*.jsonl profiles/$ROUND/ 2>/dev/null
mkdir -p $OUT/profiles && cp -r profiles/$ROUND profiles/current.json profiles/current_isa.json $OUT/profiles/
# keep the merge small: only the summaries travel back
find $OUT -name "*_agent_info.csv" -delete; find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -size +8M -delete
du -sh $OUT; tail -c 400 $OUT/bench_line.json
