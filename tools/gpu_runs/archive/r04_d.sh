#!/bin/bash
# round 4, run d: SQ counters of the row kernel (k_agg_or_rows) on configs[4], first call (no packed collection)
export TMPDIR=/tmp
O=gpurun_out/${1:-r04d}; rm -rf $O; mkdir -p $O
export BMX_GAP_PACK=0 BMX_OR_ROWS=1
B="python bench.py --config 4 --no-cpu --steps 3 --warmup 1"
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS -d $O/p1 -o x -f csv -- $B > /dev/null 2> $O/p1.err
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT -d $O/p2 -o x -f csv -- $B > /dev/null 2> $O/p2.err
timeout 600 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_WAVES GRBM_GUI_ACTIVE -d $O/p3 -o x -f csv -- $B > /dev/null 2> $O/p3.err
for p in p1 p2 p3; do
  f=$(find $O/$p -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_agg_or_rows" in r["Kernel_Name"]]
agg = collections.defaultdict(list)
for r in rows: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()): print(f"{k:28s} {sum(v)/len(v):18.0f}  (n={len(v)})")
PY
done | tee $O/summary.txt
rm -rf $O/p1 $O/p2 $O/p3
