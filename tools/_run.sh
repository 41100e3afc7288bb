mkdir -p gpurun_out
timeout 200 python tools/predict_scaling.py 8 weak > gpurun_out/predict_scaling_8_weak.jsonl 2>&1; tail -1 gpurun_out/predict_scaling_8_weak.jsonl
timeout 200 python tools/predict_scaling.py 2 weak 2>&1 | tail -1
timeout 200 python tools/predict_scaling.py 4 weak 2>&1 | tail -1
timeout 400 python tools/predict_scaling.py 8 config4 > gpurun_out/predict_scaling_8_config4.jsonl 2>&1; tail -1 gpurun_out/predict_scaling_8_config4.jsonl
