#!/bin/bash
# the slotted-slab variant of the first-call combine_or (profiles/r04_cold/slotted_slab_variant.patch, 9 % slower
# in round 4 although it requests 13 % fewer lines) against the gather form, same code base (commit ee4b344, two worktrees under
# _variants/, built here), same box, with counters: TCP / TA / TCC / SQ-LDS, one counter set per pass.
# To recreate the two trees (they are not kept):
#   git worktree add -f _variants/gather ee4b344; git worktree add -f _variants/slotted ee4b344
#   (cd _variants/slotted && git apply profiles/r04_cold/slotted_slab_variant.patch)
#   for v in gather slotted; do (cd _variants/$v && python -c 'import __graft_entry__ as g; g.build()'); done
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r05_s; rm -rf $O; mkdir -p $O
SETS=("TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" "TA_BUSY_avr TA_TA_BUSY_sum" "TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum TCC_EA_RDREQ_DRAM_sum" "TCC_TAG_STALL_sum TCC_BUSY_avr TCC_EA_RDREQ_DRAM_CREDIT_STALL_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU" "FETCH_SIZE WRITE_SIZE" "SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY GRBM_GUI_ACTIVE")
for v in gather slotted; do
  cd $R/_variants/$v
  for i in 1 2 3; do timeout 300 python bench.py --config 4 --no-cpu > $O/bench_${v}_$i.json 2>> $O/err_$v.txt; done
  for set in "${SETS[@]}"; do
    rm -rf /tmp/pmc_x
    ( cd /tmp; timeout 300 rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc_x -o x -f csv -- python $R/_variants/$v/bench.py --config 4 --no-cpu > /dev/null 2>> $O/pmc_err_$v.txt )
    python - "$(find /tmp/pmc_x -name '*counter_collection.csv' | head -1)" "k_agg_or_rows" >> $O/pmc_$v.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
try:
    for r in csv.DictReader(open(sys.argv[1])):
        if sys.argv[2] in r["Kernel_Name"]: acc[(r["Kernel_Name"][:40], r["Counter_Name"])].append(float(r["Counter_Value"]))
except Exception as e:
    print("pmc pass failed:", e)
for k, v in sorted(acc.items()): print(k[0], k[1], "per launch avg", sum(v) / len(v), "launches", len(v))
PY
  done
done
cd $R
python - <<'PY' > $O/summary.txt
import json, glob
for v in ("gather", "slotted"):
    ms = []
    for f in sorted(glob.glob("gpurun_out/r05_s/bench_%s_*.json" % v)):
        try:
            d = json.loads(open(f).read().strip().splitlines()[-1]); ms.append((d["ms_per_step"], d["roofline"].get("avg_launch_ms")))
        except Exception as e: ms.append(str(e)[:60])
    print(v, ms)
    try: print(open("gpurun_out/r05_s/pmc_%s.txt" % v).read())
    except Exception as e: print(e)
PY
cat $O/summary.txt | cut -c1-200
