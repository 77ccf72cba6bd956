mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu2.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu2.log
timeout 600 python bench.py > gpurun_out/bench2.json 2> gpurun_out/bench2.err
tail -15 gpurun_out/pytest_gpu2.log; cat gpurun_out/bench2.json
