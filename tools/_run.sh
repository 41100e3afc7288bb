mkdir -p gpurun_out
O2V_LIB=obj2voxel_amd/libobj2voxel_amd_instr.so python tools/instrument.py > gpurun_out/instr_new.json 2>&1; tail -32 gpurun_out/instr_new.json
python bench.py --no-cpu-baseline --steps 20 --warmup 3 2>&1 | tail -1 | cut -c1-900
timeout 900 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py -x -q -k "not 4096" 2>&1 | tail -8
