mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu12.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu12.log
tail -4 gpurun_out/pytest_gpu12.log
timeout 900 python tools/bench_configs.py --which 4 2>&1 | grep config | cut -c1-400
