mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu16.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu16.log
tail -3 gpurun_out/pytest_gpu16.log
timeout 900 python tools/bench_configs.py --which 5 > gpurun_out/bench_cfg5.log 2>&1; tail -3 gpurun_out/bench_cfg5.log
