mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu22.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu22.log
tail -4 gpurun_out/pytest_gpu22.log
timeout 600 python tools/bench_configs.py --which 2,4 > gpurun_out/bench_c24b.log 2>&1
grep -E '"density": "1%"' gpurun_out/bench_c24b.log | cut -c1-260 | head -8
