#!/bin/bash
# round 4, run af: the asynchronous entry (bmx_op2_dev): its tests and the configs[1] lines with materialised_async_ms_per_op
export TMPDIR=/tmp
O=gpurun_out/${1:-r04af}; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -q -m gpu -x -k "async_pairwise or pairwise or golden_case" > $O/pytest_sel.txt 2>&1; echo "rc $?" >> $O/pytest_sel.txt; tail -12 $O/pytest_sel.txt
timeout 600 python bench.py --config 1 --no-cpu > $O/bench_config1.json 2>> $O/err.txt
timeout 600 python bench.py --config 1 --density-q16 32768 --no-cpu > $O/bench_config1_50.json 2>> $O/err.txt
python - <<PY
import json
for f in ("bench_config1.json", "bench_config1_50.json"):
    j = json.loads([l for l in open("$O/" + f) if l.startswith("{")][-1])
    print(f, {k: (v["materialised_host_call_ms"], v.get("materialised_async_ms_per_op")) for k, v in j["config"]["per_op"].items()})
PY
tail -5 $O/err.txt
