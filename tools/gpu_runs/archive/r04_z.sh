#!/bin/bash
# round 4, run z: additions to the evidence pass r04final -- the two new tests, the GAP-only headline variants with their
# prepared AND-role collection, PMC of configs[4] over full-size launches only
export TMPDIR=/tmp
O=gpurun_out/${1:-r04final_b}; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -q -m gpu -k "past_2_32 or collection_budget or rank_line_memory" > $O/pytest_new.txt 2>&1; echo "rc $?" >> $O/pytest_new.txt
tail -5 $O/pytest_new.txt
for dq in 197 66; do timeout 600 python bench.py --density-q16 $dq --no-cpu --no-others --no-shard-probe > $O/bench_dq$dq.json 2>> $O/err.txt; done
rm -rf /tmp/ks; timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/ks -o s -f csv -- python bench.py --density-q16 197 --no-cpu --no-others --no-shard-probe > $O/bench_dq197_under_rocprof.json 2>> $O/err.txt
cp $(find /tmp/ks -name "*kernel_stats.csv" | head -1) $O/kernel_stats_dq197.csv
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum"; do
  rm -rf /tmp/pmc_x
  timeout 600 rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc_x -o x -f csv -- python bench.py --config 4 --no-cpu --no-subset --steps 4 --warmup 2 > /dev/null 2>> $O/pmc.err
  python - "$(find /tmp/pmc_x -name '*counter_collection.csv' | head -1)" "k_agg_or_rows|k_coll_apply" >> $O/pmc_config4.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if any(k in r["Kernel_Name"] for k in sys.argv[2].split("|")): acc[(r["Kernel_Name"][:48], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()): print(k[0], k[1], "per launch avg", sum(v) / len(v), "launches", len(v))
PY
done
cat $O/pmc_config4.txt
python - <<PY
import json
for dq in (197, 66):
    j = json.loads([l for l in open("$O/bench_dq%d.json" % dq) if l.startswith("{")][-1])
    print(dq, j["ms_per_step"], j["roofline"]["frac"], j["roofline"]["kernel"][:60], j["config"].get("prepared_collection_ms"), j["config"]["result_count"])
PY
