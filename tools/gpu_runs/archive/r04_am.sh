#!/bin/bash
# round 4, run am: collection build with 16-byte loads (count_singles / scatter / scatter_split): parity of everything packed, build time
export TMPDIR=/tmp
O=gpurun_out/${1:-r04am}; rm -rf $O; mkdir -p $O
timeout 1200 python -m pytest tests -q -m gpu -x -k "packed or collection or prepared or full_size_or or many_gap or sparse_state" > $O/pytest_sel.txt 2>&1; echo "rc $?" >> $O/pytest_sel.txt; tail -4 $O/pytest_sel.txt
timeout 600 python tools/soak_r04.py 60 B 2>&1 | tail -2
timeout 600 python bench.py --config 4 --no-cpu > $O/bench_config4.json 2>> $O/err.txt
timeout 600 python bench.py --density-q16 197 --no-cpu --no-others --no-shard-probe > $O/bench_dq197.json 2>> $O/err.txt
python - <<PY
import json
j = json.loads([l for l in open("$O/bench_config4.json") if l.startswith("{")][-1])
print("c4 cold/build/warm/break-even", j["config"]["cold_ms"], j["config"]["build_ms"], j["config"]["warm_ms"], j["config"]["break_even_calls"], j["config"]["result_count"])
j = json.loads([l for l in open("$O/bench_dq197.json") if l.startswith("{")][-1])
print("dq197", j["ms_per_step"], j["config"].get("prepared_collection_ms"), j["config"]["result_count"])
PY
