#!/bin/bash
# PMC pass over the batched-equality kernel (k_slice_eq_counts): instruction mix, LDS conflicts, wait share
export TMPDIR=/tmp
O=gpurun_out/${1:-r02o}; mkdir -p $O
cat > /tmp/eqb.py <<'PY'
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch, bitmagic_amd as bm
s = torch.cuda.Stream(); torch.cuda.set_stream(s)
ctx = bm.context(0, s.cuda_stream)
planes = [bm.bvector.generate(ctx, 0xB17A61C, 500 + i, 32768, 1_000_000_000) for i in range(32)]
rng = np.random.default_rng(3)
sc = bm.slice_scanner(ctx, planes, size=1_000_000_000)
q = [int(v) for v in rng.integers(1, 1 << 32, size=2048)]
for _ in range(3): sc.find_eq_counts(q)
PY
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY" "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  rm -rf /tmp/pmc_eq
  timeout 600 rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc_eq -o eq -f csv -- python /tmp/eqb.py > /dev/null 2>> $O/pmc.err
  f=$(find /tmp/pmc_eq -name "*counter_collection.csv" | head -1)
  python - "$f" >> $O/pmc_eq_counts.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "k_slice_eq_counts" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items(): print(k, "per launch avg", sum(v) / len(v), "launches", len(v))
PY
done
cat $O/pmc_eq_counts.txt
