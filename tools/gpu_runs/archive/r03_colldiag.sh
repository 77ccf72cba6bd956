#!/bin/bash
# where does k_coll_apply<OR,512> spend configs[4]?  tuning build: loads alone / no fold + store / both
export TMPDIR=/tmp
O=gpurun_out/${1:-r03ad}; mkdir -p $O
export BMX_LIB=$PWD/bitmagic_amd/lib/libbmx_tune.so
for d in 0 512 1024 1536; do
  BMX_DIAG_COLL=$d timeout 600 python bench.py --config 4 --steps 10 --warmup 3 --no-cpu > $O/c4_diag$d.json 2>> $O/err.txt
  BMX_DIAG_COLL=$d timeout 600 python bench.py --density-q16 197 --no-cpu --no-others --no-shard-probe --steps 10 --warmup 3 > $O/dq197_diag$d.json 2>> $O/err.txt
done
python - <<PY
import json, glob, os
for f in sorted(glob.glob("$O/*.json")):
    try:
        j = json.loads([l for l in open(f) if l.startswith("{")][-1]); print(os.path.basename(f), "ms/step", j["ms_per_step"], "kernel", j["roofline"].get("avg_launch_ms"))
    except Exception as e: print(os.path.basename(f), "FAILED", e)
PY
tail -3 $O/err.txt
