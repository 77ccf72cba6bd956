#!/bin/bash
# SQ-level PMC passes over k_coll_apply (configs[4] OR and the all-GAP AND): where do the wave cycles go?
export TMPDIR=/tmp
O=gpurun_out/${1:-r03o}; mkdir -p $O
run() { # run <tag> <bench args...>
  local tag=$1; shift
  for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS"; do
    rm -rf /tmp/pmc_y
    timeout 600 rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc_y -o y -f csv -- python bench.py "$@" > /dev/null 2>> $O/pmc.err
    python - "$(find /tmp/pmc_y -name '*counter_collection.csv' | head -1)" k_coll_apply >> $O/pmc_sq_$tag.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
try:
    for r in csv.DictReader(open(sys.argv[1])):
        if sys.argv[2] in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
except Exception as e:
    print("no counters:", e)
for k, v in sorted(acc.items()): print(k, "per launch avg", sum(v) / len(v), "launches", len(v))
PY
  done
}
run c4 --config 4 --no-cpu --steps 3 --warmup 2
run dq197 --density-q16 197 --no-cpu --no-others --no-shard-probe --steps 3 --warmup 2
cat $O/pmc_sq_c4.txt; echo ----; cat $O/pmc_sq_dq197.txt
