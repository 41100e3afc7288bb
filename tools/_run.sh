mkdir -p gpurun_out
python bench.py --no-cpu-baseline --no-capi --steps 20 --warmup 3 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['pipeline']['stages_ms'])"
python tools/bench_configs.py config 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['workload'], d['ms'], d['mvox_s'], d['stages_ms']['voxelize_ms'])"
timeout 900 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py -x -q -k "not 4096" > gpurun_out/t.log 2>&1; echo rc=$?; grep -E "passed|failed" gpurun_out/t.log
bash tools/pmc_k2.sh
