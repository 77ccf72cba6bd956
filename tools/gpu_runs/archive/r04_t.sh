#!/bin/bash
# round 4, run t: full GPU suite on the reverted (gather) row kernel + rank lines at 2x policy; bench as the driver runs it
export TMPDIR=/tmp
O=gpurun_out/${1:-r04t}; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc $?" >> $O/pytest.txt
tail -3 $O/pytest.txt
( time timeout 600 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err
tail -c 600 $O/bench_default.err
python - $O/bench_default.json <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("headline", j["value"], j["ms_per_step"], j["roofline"]["frac"])
for o in j.get("other_configs", []):
    print({k: o.get(k) for k in ("config", "density", "op", "ms_per_step", "ms_per_call", "roofline_frac", "frac", "cold_ms", "warm_ms", "build_ms", "rank_ms", "select_ms") if k in o})
PY
