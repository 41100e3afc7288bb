mkdir -p gpurun_out
timeout 60 python tools/run_workload.py cube1024 2>&1 | cut -c1-200
timeout 60 python tools/run_workload.py room2048 2>&1 | cut -c1-200
timeout 60 python tools/run_workload.py lowpoly1024 2>&1 | cut -c1-200
timeout 120 python tools/bench_configs.py config 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['workload'], d['ms'], d['mvox_s'], d['stages_ms']['voxelize_ms'])"
timeout 400 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py -x -q -k "not 4096" > gpurun_out/t.log 2>&1; echo rc=$?; grep -E "passed|failed" gpurun_out/t.log
