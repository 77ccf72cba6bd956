mkdir -p gpurun_out
timeout 900 python tools/bench_scanner.py --planes 32 --groups 512 > gpurun_out/scanner.log 2>&1
timeout 900 python tools/bench_scanner.py --planes 16 --groups 2048 >> gpurun_out/scanner.log 2>&1
cat gpurun_out/scanner.log
