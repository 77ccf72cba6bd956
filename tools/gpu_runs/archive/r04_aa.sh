#!/bin/bash
# round 4, run aa: GAP results written by k_op2_loop itself (no parked candidates, no k_emit_gaps, no layout scan): parity + configs[1] at 1 %
export TMPDIR=/tmp
O=gpurun_out/${1:-r04aa}; rm -rf $O; mkdir -p $O
timeout 1200 python -m pytest tests -q -m gpu -x -k "pairwise or op2 or stress or golden or random_block or bm64 or adapter or facade or download or clone" > $O/pytest_sel.txt 2>&1; echo "rc $?" >> $O/pytest_sel.txt
tail -4 $O/pytest_sel.txt
timeout 600 python tools/soak_r04.py 60 > $O/soak.txt 2>&1; tail -3 $O/soak.txt
timeout 600 python bench.py --config 1 --density-q16 655 --no-cpu > $O/bench_config1_1pct.json 2>> $O/err.txt
timeout 600 python bench.py --config 1 --no-cpu > $O/bench_config1.json 2>> $O/err.txt
python - <<PY
import json
for f in ("bench_config1_1pct.json", "bench_config1.json"):
    j = json.loads([l for l in open("$O/" + f) if l.startswith("{")][-1])
    print(f, {k: (v["kernel_ms"], v["materialised_host_call_ms"]) for k, v in j["config"]["per_op"].items()}, j["config"]["pair0_materialised_counts"] == j["config"]["pair0_counts_and_or_xor_sub"])
PY
