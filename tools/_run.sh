mkdir -p gpurun_out
O2V_LIB=obj2voxel_amd/libobj2voxel_amd_instr.so python tools/instrument.py 2>&1 | grep -E "wave_iter|lane_events|pushed|push onto|voxelize_ms"
python bench.py --no-cpu-baseline --no-capi --steps 20 --warmup 3 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['pipeline']['stages_ms'])"
python tools/bench_configs.py config 2>&1 | cut -c1-330
timeout 900 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py -x -q -k "not 4096" > gpurun_out/t.log 2>&1; echo rc=$?; grep -E "passed|failed" gpurun_out/t.log
