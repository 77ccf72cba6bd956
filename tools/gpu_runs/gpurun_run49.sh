mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu20.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu20.log
tail -5 gpurun_out/pytest_gpu20.log
timeout 600 python tools/bench_small.py > gpurun_out/bench_small.log 2>&1; tail -3 gpurun_out/bench_small.log
