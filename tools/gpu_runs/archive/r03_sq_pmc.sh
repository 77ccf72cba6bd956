#!/bin/bash
# SQ-level PMC passes over the kernels new in round 3: k_count_op2_loop (configs[1] at 1 %), k_rank_lines / k_select_sdir (configs[3])
export TMPDIR=/tmp
O=gpurun_out/${1:-r03aj}; mkdir -p $O
run() { # run <tag> <kernel substrings a|b> <bench args...>
  local tag=$1 kern=$2; shift 2
  for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS"; do
    rm -rf /tmp/pmc_y
    timeout 300 rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc_y -o y -f csv -- python bench.py "$@" > /dev/null 2>> $O/pmc.err
    python - "$(find /tmp/pmc_y -name '*counter_collection.csv' | head -1)" "$kern" >> $O/pmc_sq_$tag.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
try:
    for r in csv.DictReader(open(sys.argv[1])):
        for k in sys.argv[2].split("|"):
            if k in r["Kernel_Name"]: acc[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
except Exception as e:
    print("no counters:", e)
for k, v in sorted(acc.items()): print(k[0], k[1], "per launch avg", sum(v) / len(v), "launches", len(v))
PY
  done
}
run c1_1pct "k_count_op2_loop" --config 1 --density-q16 655 --no-cpu --steps 3 --warmup 1
run c3 "k_rank_lines|k_select_sdir" --config 3 --no-cpu --steps 3 --warmup 1
cat $O/pmc_sq_c1_1pct.txt; echo ----; cat $O/pmc_sq_c3.txt; tail -2 $O/pmc.err
