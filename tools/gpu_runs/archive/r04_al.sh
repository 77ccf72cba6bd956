#!/bin/bash
# round 4, run al: configs[1] at 1 % and 10 % with the asynchronous entry over any block kinds; the whole GPU suite
export TMPDIR=/tmp
O=gpurun_out/${1:-r04al}; rm -rf $O; mkdir -p $O
timeout 600 python bench.py --config 1 --density-q16 655 --no-cpu > $O/bench_config1_1pct.json 2>> $O/err.txt
timeout 600 python bench.py --config 1 --no-cpu > $O/bench_config1.json 2>> $O/err.txt
python - <<PY
import json
for f in ("bench_config1_1pct.json", "bench_config1.json"):
    j = json.loads([l for l in open("$O/" + f) if l.startswith("{")][-1])
    print(f, {k: (v["materialised_host_call_ms"], v.get("materialised_async_ms_per_op")) for k, v in j["config"]["per_op"].items()})
PY
timeout 1500 python -m pytest tests -q -m gpu > $O/pytest.txt 2>&1; echo "rc $?" >> $O/pytest.txt; grep -E "passed|failed" $O/pytest.txt
tail -3 $O/err.txt
