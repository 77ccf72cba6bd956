#!/bin/bash
# round 5: crossover of k_agg_and_rows against the wave-per-item kernels by operands per group
export TMPDIR=/tmp
O=gpurun_out/r05_b; rm -rf $O; mkdir -p $O
make -s -C bitmagic_amd/csrc >/dev/null 2>&1
CROSS=1 DQS=66,197 timeout 900 python tools/bench_and_rows.py 2> $O/err.txt | grep ops_per_group > $O/cross.jsonl
python - <<PY
import json
for l in open("$O/cross.jsonl"):
    r = json.loads(l); print(r["dq"], r["ops_per_group"], r["groups"], r["alg_GB"], "ipw4", r["rows128_ms"], "ipw1", r["rows_ms"], "ipw8", r["rows512_ms"], "older", r["older_ms"], r["rows_sum"] == r["older_sum"])
PY
