#!/bin/bash
# Round 6: SQ counters of k_count_op2_loop on the mixed 1 % pair and (BMX_PAIR_STREAM=0) on the all-bit 10 % pair -- what the
# same kernel does differently when a third of the blocks are GAP (no stall / occupancy counters for it).
export TMPDIR=/tmp
O=gpurun_out/r06_pair; mkdir -p $O; R=$PWD
for cfg in "655 -1 mixed_1pct" "6554 0 bitblocks_10pct"; do
  set -- $cfg
  for pmc in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD" "GRBM_GUI_ACTIVE"; do
    rm -rf /tmp/pmc_y
    ( cd /tmp; BMX_PAIR_STREAM=$2 timeout 300 rocprofv3 --kernel-trace --pmc $pmc -d /tmp/pmc_y -o y -f csv -- python $R/bench.py --config 1 --density-q16 $1 --no-cpu --steps 3 --warmup 1 > /dev/null 2>> $R/$O/pmc.err )
    python - "$(find /tmp/pmc_y -name '*counter_collection.csv' | head -1)" $3 >> $O/sq_count_op2_loop.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
try:
    for r in csv.DictReader(open(sys.argv[1])):
        if "k_count_op2_loop" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
except Exception as e:
    print("pmc pass failed:", e)
for k, v in sorted(acc.items()): print(sys.argv[2], "k_count_op2_loop<4,true>", k, "per launch avg", sum(v) / len(v), "launches", len(v))
PY
  done
done
cat $O/sq_count_op2_loop.txt
