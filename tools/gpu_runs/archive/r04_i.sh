#!/bin/bash
# round 4, run i: bench --config 4 with cold / build / warm kept apart
export TMPDIR=/tmp
O=gpurun_out/${1:-r04i}; mkdir -p $O
timeout 900 python bench.py --config 4 --steps 10 --warmup 2 > $O/bench_config4.json 2> $O/bench_config4.err
python - <<PY
import json
j = json.loads([l for l in open("$O/bench_config4.json") if l.startswith("{")][-1])
c = j["config"]
print({k: c[k] for k in ("cold_ms", "build_ms", "prepare_call_wall_ms", "warm_ms", "warm_count_equal", "break_even_calls", "subset_of_the_collection", "result_count")})
print(j["ms_per_step"], j["roofline"]["frac"], j["roofline"]["warm"]["frac"], j.get("cpu_baseline"))
PY
tail -3 $O/bench_config4.err
