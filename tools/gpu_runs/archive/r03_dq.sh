export TMPDIR=/tmp
O=gpurun_out/r03am; mkdir -p $O
for dq in 655 460 328; do timeout 600 python bench.py --density-q16 $dq --no-cpu --no-others --no-shard-probe --steps 10 --warmup 3 > $O/bench_dq$dq.json 2>> $O/err.txt; done
python - <<PY
import json, glob, os
for f in sorted(glob.glob("$O/bench_dq*.json")):
    j = json.loads([l for l in open(f) if l.startswith("{")][-1]); print(os.path.basename(f), "ms/step", j["ms_per_step"], "frac", j["roofline"]["frac"], j["roofline"]["kernel"][:50], j["config"]["block_types_vec0"], "GB", round(j["roofline"]["algorithmic_bytes_per_step"]/1e9, 2))
PY
