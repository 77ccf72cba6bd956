mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu3.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu3.log
timeout 600 python bench.py > gpurun_out/bench3.json 2> gpurun_out/bench3.err
tail -12 gpurun_out/pytest_gpu3.log; python -c "
import json;d=json.load(open('gpurun_out/bench3.json'));print(d['value'],d['ms_per_step'],d['roofline'],d['cpu_baseline'])"
