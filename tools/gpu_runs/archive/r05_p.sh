#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r05_p; rm -rf $O; mkdir -p $O
COMMON=0 VARS=rows_256_3_nt0,filter_256_3,filter_256_4 timeout 600 python tools/bench_and_rows.py > $O/and_rows_independent.jsonl 2> $O/err.txt
python - <<'PY' > $O/summary.txt
import json
for l in open('gpurun_out/r05_p/and_rows_independent.jsonl'):
    d=json.loads(l); print(d['dq'], d['variant'], d['ms'], d['frac'], d['count'], d['count_ok'], d['alg_GB'])
PY
cat $O/summary.txt
# (record of the run behind profiles/r05_and_rows/filter_form/and_rows_independent.jsonl: needs that directory's patch for the filter_* variants)
