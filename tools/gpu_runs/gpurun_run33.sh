mkdir -p gpurun_out
timeout 600 python tools/bench_upload.py > gpurun_out/upload.log 2>&1; cat gpurun_out/upload.log | grep path
