mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu13.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu13.log
tail -3 gpurun_out/pytest_gpu13.log
timeout 900 python tools/bench_configs.py --which 2 2>&1 | grep '"1%"' | cut -c1-330
