for lib in libobj2voxel_amd.so libvar_w5.so; do echo "== $lib"; 
O2V_LIB=obj2voxel_amd/$lib timeout 100 python bench.py --no-cpu-baseline --no-capi --steps 20 --warmup 3 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['pipeline']['stages_ms'])"
O2V_LIB=obj2voxel_amd/$lib timeout 60 python tools/run_workload.py config2_blend 2>&1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['workload'], d['ms'], d['stages_ms'])"
done
timeout 300 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_parity.py -x -q > gpurun_out/t.log 2>&1; echo rc=$?; grep -E "passed|failed" gpurun_out/t.log
O2V_LIB=obj2voxel_amd/libvar_w5.so timeout 300 python -m pytest tests/test_gpu_fuzz.py -x -q > gpurun_out/t5.log 2>&1; echo rc=$?; grep -E "passed|failed" gpurun_out/t5.log
