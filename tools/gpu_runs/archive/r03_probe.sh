#!/bin/bash
# materialised pairwise ops: load / store cache policy and workgroups per CU of k_op2_stream, next to torch's plain 2R:1W kernel
export TMPDIR=/tmp
O=gpurun_out/${1:-r03ab}; mkdir -p $O
for nt in 3 2 1 0; do for w in 2 4 8; do
  BMX_OP2_NT=$nt BMX_OP2_WGS=$w timeout 300 python bench.py --config 1 --no-cpu --steps 5 --warmup 2 > $O/c1_nt${nt}_w$w.json 2>> $O/err.txt
done; done
BMX_PAIR_STREAM=0 timeout 300 python bench.py --config 1 --no-cpu --steps 5 --warmup 2 > $O/c1_percolumn.json 2>> $O/err.txt
python - <<PY
import json, glob, os
for f in sorted(glob.glob("$O/c1_*.json")):
    j = json.loads([l for l in open(f) if l.startswith("{")][-1])
    print(os.path.basename(f), {k: v["materialised_host_call_ms"] for k, v in j["config"]["per_op"].items()}, (j["config"]["read_write_probe"] or {}).get("ms"))
PY
tail -2 $O/err.txt
