#!/bin/bash
# round 4, run r: the headline bench as the driver runs it (with the other configs' summaries) + --config 0
export TMPDIR=/tmp
O=gpurun_out/${1:-r04r}; mkdir -p $O
( time timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err ) 2> $O/time.txt
timeout 300 python bench.py --config 0 > $O/bench_config0.json 2>> $O/bench.err
python - <<PY
import json
j = json.loads([l for l in open("$O/bench.json") if l.startswith("{")][-1])
print("headline", j["value"], j["ms_per_step"], j["roofline"]["frac"], j.get("cpu_baseline", {}).get("matches_gpu_full"))
for k, v in j.get("other_configs", {}).items():
    if "error" in v: print(k, "ERROR", v["error"]); continue
    print(k, v["ms_per_step"], v["roofline"].get("frac"), v.get("cpu_baseline"), v.get("wall_s"), {x: v.get(x) for x in ("cold_ms", "build_ms", "warm_ms") if x in v})
j0 = json.loads([l for l in open("$O/bench_config0.json") if l.startswith("{")][-1])
print("config0", j0["value"], j0["ms_per_step"], j0["config"]["counts_equal"], j0["cpu_baseline"])
PY
cat $O/time.txt; tail -3 $O/bench.err
