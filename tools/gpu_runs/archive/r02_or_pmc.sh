#!/bin/bash
# PMC passes over the column-tile OR kernel (configs[4]) for two variants
export TMPDIR=/tmp
O=gpurun_out/${1:-r02f}; rm -rf $O; mkdir -p $O
for v in 0; do
  export BMX_OR_TILE=$v
  timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS -d $O/p1_$v -o x -f csv -- python bench.py --config 4 --no-cpu --steps 3 --warmup 1 > /dev/null 2> $O/p1_$v.err
  timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT -d $O/p2_$v -o x -f csv -- python bench.py --config 4 --no-cpu --steps 3 --warmup 1 > /dev/null 2> $O/p2_$v.err
  timeout 600 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_WAVES GRBM_GUI_ACTIVE -d $O/p3_$v -o x -f csv -- python bench.py --config 4 --no-cpu --steps 3 --warmup 1 > /dev/null 2> $O/p3_$v.err
done
python - <<'PY'
import csv, glob, collections, os, sys
O = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r02f"
PY
for v in 0; do for p in p1 p2 p3; do
  f=$(find $O/${p}_$v -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" "$v" <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_agg_or_gap" in r["Kernel_Name"]]
agg = collections.defaultdict(list)
for r in rows: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()): print(f"or_tile={sys.argv[2]} {k:28s} {sum(v)/len(v):18.0f}  (n={len(v)})")
PY
done; done | tee $O/summary.txt
