#!/bin/bash
# end of round: the driver's three commands with the library as committed (GPU tests, smoke, the default bench line)
export TMPDIR=/tmp
O=gpurun_out/${1:-r03final}; mkdir -p $O
( time timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1 ) 2> $O/summary.txt; echo "pytest gpu rc=$?" >> $O/summary.txt
tail -3 $O/pytest.txt >> $O/summary.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?" >> $O/summary.txt
( time timeout 900 python bench.py > $O/bench.json 2> $O/bench.err ) 2>> $O/summary.txt
python - <<PY >> $O/summary.txt
import json
j = json.loads([l for l in open("$O/bench.json") if l.startswith("{")][-1])
print("bench", j["value"], j["unit"], "ms/step", j["ms_per_step"], "frac", j["roofline"]["frac"], "cpu", j["cpu_baseline"]["value"], j["cpu_baseline"].get("matches_gpu_full"))
for k, v in j["other_configs"].items(): print(k, v["ms_per_step"], v["roofline"]["frac"], {a: b for a, b in v["cpu_baseline"].items() if a.startswith("match")})
PY
cat $O/summary.txt
