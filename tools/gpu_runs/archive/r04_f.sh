#!/bin/bash
# round 4, run f: HBM traffic of the row kernel (FETCH_SIZE / WRITE_SIZE / TCC counters in separate passes) on configs[4], first call
export TMPDIR=/tmp
O=gpurun_out/${1:-r04f}; rm -rf $O; mkdir -p $O
export BMX_GAP_PACK=0 BMX_OR_ROWS=1
B="python bench.py --config 4 --no-cpu --steps 3 --warmup 1"
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/p1 -o x -f csv -- $B > /dev/null 2> $O/p1.err
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/p2 -o x -f csv -- $B > /dev/null 2> $O/p2.err
timeout 600 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum -d $O/p3 -o x -f csv -- $B > /dev/null 2> $O/p3.err
timeout 600 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum -d $O/p4 -o x -f csv -- $B > /dev/null 2> $O/p4.err
for p in p1 p2 p3 p4; do
  f=$(find $O/$p -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_agg_or_rows" in r["Kernel_Name"]]
agg = collections.defaultdict(list)
for r in rows: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()): print(f"{k:28s} {sum(v)/len(v):18.0f}  (n={len(v)})")
PY
done | tee $O/summary.txt
tail -2 $O/p*.err | tail -12
rm -rf $O/p1 $O/p2 $O/p3 $O/p4
