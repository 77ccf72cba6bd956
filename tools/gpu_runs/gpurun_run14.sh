mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu4.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu4.log
tail -12 gpurun_out/pytest_gpu4.log
timeout 900 python tools/bench_configs.py --which 5 --or-vecs 512 > gpurun_out/config5_512.log 2>&1; tail -3 gpurun_out/config5_512.log
timeout 900 python tools/bench_configs.py --which 5 --or-vecs 4096 > gpurun_out/config5_4096.log 2>&1; tail -3 gpurun_out/config5_4096.log
