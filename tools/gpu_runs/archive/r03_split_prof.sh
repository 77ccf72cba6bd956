#!/bin/bash
# configs[4] with split bags: bench line, rocprofv3 --stats, HBM traffic (FETCH_SIZE / WRITE_SIZE / TCC) of k_coll_apply
export TMPDIR=/tmp
O=gpurun_out/${1:-r03aq}; mkdir -p $O
timeout 600 python bench.py --config 4 > $O/bench_config4.json 2>> $O/err.txt
rm -rf /tmp/ks; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/ks -o c4 -f csv -- python bench.py --config 4 --no-cpu > $O/bench_config4_under_rocprof.json 2>> $O/err.txt
cp $(find /tmp/ks -name "*kernel_stats.csv" | head -1) $O/kernel_stats_config4.csv
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  rm -rf /tmp/pmc_x
  timeout 600 rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc_x -o x -f csv -- python bench.py --config 4 --no-cpu --steps 4 --warmup 2 > /dev/null 2>> $O/pmc.err
  python - "$(find /tmp/pmc_x -name '*counter_collection.csv' | head -1)" >> $O/pmc_config4.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "k_coll_apply" in r["Kernel_Name"]: acc[(r["Kernel_Name"][:48], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()): print(k[0], k[1], "per launch avg", sum(v) / len(v), "launches", len(v))
PY
done
python - <<PY
import json
for f in ("bench_config4.json", "bench_config4_under_rocprof.json"):
    j = json.loads([l for l in open("$O/" + f) if l.startswith("{")][-1]); r = j["roofline"]
    print(f, "ms/step", j["ms_per_step"], "achieved", r["achieved"], "frac", r["frac"], "ref_fmt", r["reference_format_GBps"], "alg", r["algorithmic_bytes_per_launch"], j["config"]["packed_collection"], j.get("cpu_baseline", {}).get("matches_gpu_sample"))
PY
grep k_coll $O/kernel_stats_config4.csv | cut -c1-60,150-260; cat $O/pmc_config4.txt
