export TMPDIR=/tmp
O=gpurun_out/r03final3; mkdir -p $O
( time timeout 900 python bench.py > $O/bench.json 2> $O/bench.err ) 2> $O/summary.txt
python - <<PY >> $O/summary.txt
import json
j = json.loads([l for l in open("$O/bench.json") if l.startswith("{")][-1])
print("bench", j["value"], j["unit"], "ms/step", j["ms_per_step"], "frac", j["roofline"]["frac"], "cpu", j["cpu_baseline"]["value"], j["cpu_baseline"].get("matches_gpu_full"))
for k, v in j["other_configs"].items(): print(k, "ms/step", v["ms_per_step"], "steps", v.get("steps"), "frac", v["roofline"]["frac"], {a: b for a, b in v["cpu_baseline"].items() if a.startswith("match")}, v.get("rank_ms"), v.get("select_ms"))
PY
cat $O/summary.txt
