mkdir -p gpurun_out
timeout 900 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final_smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/final_smoke.log
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/final_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/final_pytest.log
timeout 600 python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu > gpurun_out/final_bench_dist.json 2> gpurun_out/final_bench_dist.err
./tests/cpp/_bin/test_facade > gpurun_out/cpp_tests.log 2>&1; echo "facade rc=$?" >> gpurun_out/cpp_tests.log
./oracle/_ref/test_adapter_ref >> gpurun_out/cpp_tests.log 2>&1; echo "adapter rc=$?" >> gpurun_out/cpp_tests.log
tail -2 gpurun_out/final_smoke.log; tail -3 gpurun_out/final_pytest.log; tail -4 gpurun_out/cpp_tests.log
python -c "
import json
for f in ('final_bench.json','final_bench_dist.json'):
    d=json.loads(open('gpurun_out/'+f).read().strip().splitlines()[-1]);print(f, d['value'],d['ms_per_step'],d['roofline']['frac'],d['roofline']['traffic'],d.get('cpu_baseline'))"
