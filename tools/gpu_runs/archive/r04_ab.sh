#!/bin/bash
# round 4, run ab: configs[1] at 1 % with k_op2_loop laying its GAP candidates out (bench + rocprofv3 --stats), pairwise / op2 tests, soak part C
export TMPDIR=/tmp
O=gpurun_out/${1:-r04ab}; rm -rf $O; mkdir -p $O
timeout 600 python bench.py --config 1 --density-q16 655 --no-cpu > $O/bench_c1_1pct_plain.json 2>> $O/err.txt
rm -rf /tmp/ks; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/ks -o s -f csv -- python bench.py --config 1 --density-q16 655 --no-cpu > $O/bench_c1_1pct.json 2>> $O/err.txt
cp $(find /tmp/ks -name "*kernel_stats.csv" | head -1) $O/kernel_stats_c1_1pct.csv
python - <<PY
import csv, json
for r in csv.DictReader(open("$O/kernel_stats_c1_1pct.csv")):
    if any(k in r["Name"] for k in ("k_op2_loop","k_move_gaps","k_rebase","copyBuffer","k_scan_layout","k_emit_gaps","k_probe_rw")):
        print(r["Name"][:40], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
j=json.loads([l for l in open("$O/bench_c1_1pct_plain.json") if l.startswith("{")][-1])
print({k:(v['kernel_ms'],v['materialised_host_call_ms']) for k,v in j['config']['per_op'].items()})
PY
timeout 900 python -m pytest tests -q -m gpu -x -k "pairwise or op2 or stress or golden or random_block" 2>&1 | tail -2
timeout 600 python tools/soak_r04.py 30 2>&1 | tail -2
