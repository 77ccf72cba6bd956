#!/bin/bash
# PMC passes (FETCH_SIZE, TCC) over the pairwise streaming count kernel of bench.py --config 1
export TMPDIR=/tmp
O=gpurun_out/${1:-r02t}; mkdir -p $O
for set in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  rm -rf /tmp/pmc_c1
  timeout 600 rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc_c1 -o c1 -f csv -- python bench.py --config 1 --no-cpu --steps 5 --warmup 1 > /dev/null 2>> $O/pmc.err
  f=$(find /tmp/pmc_c1 -name "*counter_collection.csv" | head -1)
  python - "$f" >> $O/pmc_config1.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "k_count_op2_stream" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items(): print(k, "per launch avg", sum(v) / len(v), "launches", len(v))
PY
done
cat $O/pmc_config1.txt
