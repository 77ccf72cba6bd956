#!/bin/bash
# round 4, run q: lines per select query past the L2 against the directory granularity
export TMPDIR=/tmp
O=gpurun_out/${1:-r04q}; rm -rf $O; mkdir -p $O
for sh in 0 12 14; do
  BMX_RS_SDIR_SHIFT=$sh timeout 600 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum -d $O/p$sh -o x -f csv -- python bench.py --config 3 --no-cpu --steps 3 --warmup 1 > /dev/null 2>> $O/err.txt
  f=$(find $O/p$sh -name "*counter_collection.csv" | head -1)
  python - "$f" $sh <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_select_sdir" in r["Kernel_Name"]]
agg = collections.defaultdict(list)
for r in rows: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("shift", sys.argv[2], {k: round(sum(v)/len(v)/1e6, 2) for k, v in sorted(agg.items())}, "M per 10 M queries")
PY
  rm -rf $O/p$sh
done | tee $O/summary.txt
