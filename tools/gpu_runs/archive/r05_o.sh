#!/bin/bash
# the filter form of the first-call AND over GAP-only operands (k_agg_and_filter): parity, then timing against the union form
export TMPDIR=/tmp
O=gpurun_out/r05_o; rm -rf $O; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "and_rows or search_count or counting_formulation or launch_shape" ) > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/summary.txt
tail -5 $O/pytest.txt >> $O/summary.txt
timeout 600 python tools/soak_r05.py 60 F > $O/soak_r05.txt 2>&1; echo "soak_r05 rc=$?" >> $O/summary.txt; tail -2 $O/soak_r05.txt >> $O/summary.txt
VARS=gapcount_r4,rows_256_3_nt0,filter_256_2,filter_256_3,filter_256_4 timeout 600 python tools/bench_and_rows.py > $O/and_rows.jsonl 2> $O/and_rows.err; echo "bench rc=$?" >> $O/summary.txt
python - <<'PY' >> $O/summary.txt
import json
for l in open('gpurun_out/r05_o/and_rows.jsonl'):
    d=json.loads(l); print(d['dq'], d['variant'], d['ms'], d['frac'], d['count_ok'], d['kernel'][:40], d['materialised_host_ms'])
PY
cat $O/summary.txt
# (record of the run behind profiles/r05_and_rows/filter_form: the filter_* variants exist only with that directory's patch applied)
