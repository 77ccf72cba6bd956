#!/bin/bash
# round 3, pass C: rank lanes-per-query sweep, packed-collection tests again, kernel stats + PMC of the packed kernels
export TMPDIR=/tmp
O=gpurun_out/${1:-r03c}; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "packed or rank_select_queries or full_size_rank" > $O/pytest_new.txt 2>&1; echo "pytest new rc=$?" >> $O/summary.txt
tail -3 $O/pytest_new.txt >> $O/summary.txt
for l in 8 4 2; do
  BMX_RS_LANES=$l timeout 600 python bench.py --config 3 --no-cpu --steps 10 > $O/bench_config3_l$l.json 2>> $O/err.txt
  BMX_RS_LANES=$l timeout 600 python bench.py --config 3 --no-cpu --steps 10 --density-q16 655 > $O/bench_config3_1pct_l$l.json 2>> $O/err.txt
done
# kernel stats of config 4 and the all-GAP AND through the packed collections
rm -rf /tmp/ks4; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/ks4 -o c4 -f csv -- python bench.py --config 4 --steps 10 --warmup 3 --no-cpu > $O/bench_config4.json 2>> $O/err.txt
cp $(find /tmp/ks4 -name "*kernel_stats.csv" | head -1) $O/kernel_stats_config4.csv
rm -rf /tmp/ks5; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/ks5 -o d197 -f csv -- python bench.py --density-q16 197 --no-cpu --no-others --no-shard-probe --steps 10 --warmup 3 > $O/bench_dq197.json 2>> $O/err.txt
cp $(find /tmp/ks5 -name "*kernel_stats.csv" | head -1) $O/kernel_stats_dq197.csv
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_MISS_sum"; do
  rm -rf /tmp/pmc4
  timeout 600 rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc4 -o c4 -f csv -- python bench.py --config 4 --steps 4 --warmup 2 --no-cpu > /dev/null 2>> $O/pmc.err
  python - "$(find /tmp/pmc4 -name '*counter_collection.csv' | head -1)" k_coll_apply >> $O/pmc_config4.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Kernel_Name"]: acc[(r["Kernel_Name"][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()): print(k[0], k[1], "per launch avg", sum(v) / len(v), "launches", len(v))
PY
  rm -rf /tmp/pmc5
  timeout 600 rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc5 -o d -f csv -- python bench.py --density-q16 197 --no-cpu --no-others --no-shard-probe --steps 4 --warmup 2 > /dev/null 2>> $O/pmc.err
  python - "$(find /tmp/pmc5 -name '*counter_collection.csv' | head -1)" k_coll_apply >> $O/pmc_dq197.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Kernel_Name"]: acc[(r["Kernel_Name"][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()): print(k[0], k[1], "per launch avg", sum(v) / len(v), "launches", len(v))
PY
done
python - <<'PY' >> $O/summary.txt
import json, glob, os
O = os.environ.get("O", "gpurun_out/r03c")
for f in sorted(glob.glob(O + "/bench_*.json")):
    try:
        j = json.loads([l for l in open(f) if l.startswith("{")][-1])
        c = j["config"]
        extra = {k: c.get(k) for k in ("rank_ms", "select_ms", "result_count", "packed_collection") if k in c}
        print(os.path.basename(f), "ms/step", j["ms_per_step"], "frac", j["roofline"].get("frac"), "kernel", str(j["roofline"].get("kernel"))[:50], extra)
    except Exception as e:
        print(os.path.basename(f), "FAILED", e)
PY
cat $O/summary.txt; head -8 $O/kernel_stats_config4.csv; head -8 $O/kernel_stats_dq197.csv; cat $O/pmc_config4.txt $O/pmc_dq197.txt
