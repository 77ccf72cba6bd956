mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu6.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu6.log
tail -4 gpurun_out/pytest_gpu6.log
timeout 900 python tools/bench_configs.py --which 2 > gpurun_out/configs2.log 2>&1; grep -E '"10%"|"1%"' gpurun_out/configs2.log | cut -c1-330
