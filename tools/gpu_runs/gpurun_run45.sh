mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu18.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu18.log
tail -12 gpurun_out/pytest_gpu18.log
