mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu5.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu5.log
tail -4 gpurun_out/pytest_gpu5.log
timeout 900 python tools/bench_configs.py --which 2,4,5 > gpurun_out/configs245.log 2>&1; cat gpurun_out/configs245.log | cut -c1-420
