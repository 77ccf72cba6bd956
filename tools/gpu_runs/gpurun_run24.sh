mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu8.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu8.log
tail -6 gpurun_out/pytest_gpu8.log
timeout 900 python tools/bench_scanner.py --planes 32 --groups 512 > gpurun_out/scanner2.log 2>&1
timeout 900 python tools/bench_scanner.py --planes 16 --groups 2048 >> gpurun_out/scanner2.log 2>&1
grep pattern gpurun_out/scanner2.log | cut -c1-260; tail -3 gpurun_out/scanner2.log | grep -v pattern
