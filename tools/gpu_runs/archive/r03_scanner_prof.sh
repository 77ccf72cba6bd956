#!/bin/bash
# rocprofv3 evidence for the scanner kernels: --stats of tools/bench_scanner.py, then FETCH_SIZE / TCC_MISS of the equality-count and comparison kernels
export TMPDIR=/tmp
O=gpurun_out/${1:-r03ao}; mkdir -p $O
rm -rf /tmp/ks; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/ks -o s -f csv -- python tools/bench_scanner.py > $O/bench_scanner_under_rocprof.log 2>> $O/err.txt
cp $(find /tmp/ks -name "*kernel_stats.csv" | head -1) $O/kernel_stats_scanner.csv
for set in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  rm -rf /tmp/pmc_s
  timeout 600 rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc_s -o s -f csv -- python tools/bench_scanner.py > /dev/null 2>> $O/pmc.err
  python - "$(find /tmp/pmc_s -name '*counter_collection.csv' | head -1)" >> $O/pmc_scanner.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    for k in ("k_slice_eq_counts", "k_slice_compare", "k_pipe_counts_staged"):
        if k in r["Kernel_Name"]: acc[(r["Kernel_Name"][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()): print(k[0], k[1], "per launch avg", sum(v) / len(v), "min", min(v), "max", max(v), "launches", len(v))
PY
done
head -12 $O/kernel_stats_scanner.csv | cut -c1-220; cat $O/pmc_scanner.txt | cut -c1-200
