#!/bin/bash
# round 5, first pass: the new AND rows kernel -- parity tests, then launch shapes against the round-4 kernels, then the
# timing probes of the tuning build (loads alone / no fold)
export TMPDIR=/tmp
O=gpurun_out/r05_a; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "and_rows or counting_formulation or many_gap or uploaded_only or sparse_state or golden_case or packed or prepared" > $O/pytest_sel.txt 2>&1; echo "pytest rc=$?" >> $O/summary.txt
tail -5 $O/pytest_sel.txt >> $O/summary.txt
timeout 900 python tools/bench_and_rows.py > $O/and_rows.jsonl 2> $O/and_rows.err; echo "bench rc=$?" >> $O/summary.txt
BMX_LIB=$PWD/bitmagic_amd/lib/libbmx_tune.so timeout 900 python tools/bench_and_rows.py > $O/and_rows_diag.jsonl 2> $O/and_rows_diag.err; echo "diag rc=$?" >> $O/summary.txt
python - <<PY >> $O/summary.txt
import json
for f in ("$O/and_rows.jsonl", "$O/and_rows_diag.jsonl"):
    for l in open(f):
        r = json.loads(l); print(r["dq"], r["variant"], r["ms"], r["frac"], r["count_ok"], r["materialised_host_ms"])
PY
