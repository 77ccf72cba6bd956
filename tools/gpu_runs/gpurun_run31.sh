mkdir -p gpurun_out
for dq in 655 66; do
timeout 600 python bench.py --density-q16 $dq --no-cpu --steps 10 --warmup 2 > gpurun_out/bench_dq$dq.json 2> gpurun_out/bench_dq$dq.err
python -c "
import json;d=json.load(open('gpurun_out/bench_dq$dq.json'));print($dq, d['ms_per_step'], d['value'], d['config']['block_types_vec0'], d['roofline']['algorithmic_bytes_per_launch'], d['roofline']['achieved'], d['config']['result_count'])"
done
