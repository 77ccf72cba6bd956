mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu10.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu10.log
tail -5 gpurun_out/pytest_gpu10.log
timeout 600 python bench.py > gpurun_out/bench4.json 2> gpurun_out/bench4.err
python -c "
import json;d=json.load(open('gpurun_out/bench4.json'));print(d['value'],d['ms_per_step'],d['roofline']['frac'],d['cpu_baseline']['value'],d['cpu_baseline']['matches_gpu'])"
