mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu14.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu14.log
tail -3 gpurun_out/pytest_gpu14.log
for arg in "--density-q16 328" "--density-q16 200"; do
timeout 600 python bench.py $arg --no-cpu --steps 10 --warmup 2 > gpurun_out/bench_tmp.json 2> gpurun_out/bench_tmp.err
python -c "
import json;d=json.load(open('gpurun_out/bench_tmp.json'));print('$arg', d['ms_per_step'], d['config']['block_types_vec0'], round(d['roofline']['achieved']), d['config']['result_count'])"
done
