mkdir -p gpurun_out
timeout 900 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final_smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/final_smoke.log
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/final_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/final_pytest.log
timeout 600 python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err
tail -2 gpurun_out/final_smoke.log; tail -3 gpurun_out/final_pytest.log; python -c "
import json;d=json.load(open('gpurun_out/final_bench.json'));print(d['value'],d['ms_per_step'],d['roofline']['frac'],d['roofline']['traffic'],d['cpu_baseline']['value'],d['cpu_baseline']['kind'],d['cpu_baseline']['matches_gpu'])"
