mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_cpp_facade.py -m gpu -x -q > gpurun_out/pytest_cpp.log 2>&1; echo "rc=$?" >> gpurun_out/pytest_cpp.log
timeout 900 python tools/bench_configs.py --which 2,4 > gpurun_out/configs24.log 2>&1
tail -5 gpurun_out/pytest_cpp.log; cat gpurun_out/configs24.log
