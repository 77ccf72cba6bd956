#!/bin/bash
# round 4: refresh of the evidence pass after the last library change (k_op2_loop lays its GAP candidates out itself): the
# driver's three commands, and the lines / profiles the change touches
export TMPDIR=/tmp
O=gpurun_out/${1:-r04final_c}; rm -rf $O; mkdir -p $O
( time timeout 2400 python -m pytest tests -q -m gpu > $O/pytest.txt 2>&1 ) 2>> $O/summary.txt; echo "pytest gpu rc=$?" >> $O/summary.txt
grep -E "passed|failed" $O/pytest.txt >> $O/summary.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?" >> $O/summary.txt
( time timeout 900 python bench.py > $O/bench.json 2> $O/bench.err ) 2>> $O/summary.txt
timeout 600 python bench.py --config 1 --density-q16 655 > $O/bench_config1_1pct.json 2>> $O/err.txt
timeout 600 python bench.py --config 4 > $O/bench_config4.json 2>> $O/err.txt
stats_of() { local name=$1; shift
  rm -rf /tmp/ks; timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/ks -o s -f csv -- python bench.py "$@" > $O/bench_${name}_under_rocprof.json 2>> $O/err.txt
  cp $(find /tmp/ks -name "*kernel_stats.csv" | head -1) $O/kernel_stats_$name.csv; }
stats_of config1_1pct --config 1 --density-q16 655 --no-cpu
stats_of config4 --config 4 --no-cpu --no-subset
timeout 600 python tools/op2_ab.py 655 2>/dev/null > $O/op2_1pct.json
timeout 900 python tools/soak_r04.py 120 > $O/soak_r04.log 2>&1
tail -2 $O/soak_r04.log >> $O/summary.txt
python - <<PY >> $O/summary.txt
import json, glob, os
for f in sorted(glob.glob("$O/bench*.json")):
    try:
        j = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(os.path.basename(f), j["value"], j["unit"], "ms/step", j["ms_per_step"], "frac", j["roofline"].get("frac"), str(j["roofline"].get("kernel"))[:46])
    except Exception as e:
        print(os.path.basename(f), "FAILED", e)
PY
cat $O/summary.txt; cat $O/op2_1pct.json
