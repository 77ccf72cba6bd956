#!/bin/bash
# packed collections: split bags (single-bit runs as 16-bit positions) on / off
export TMPDIR=/tmp
O=gpurun_out/${1:-r03ap}; mkdir -p $O
for sp in 1 0; do BMX_COLL_SPLIT=$sp timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_group.py -x -q -m gpu -k "packed or many_gap or 4096 or sparse or weighted or hint" > $O/pytest_split$sp.txt 2>&1; tail -2 $O/pytest_split$sp.txt; done
for sp in 1 0 1; do
  BMX_COLL_SPLIT=$sp timeout 600 python bench.py --config 4 --steps 10 --warmup 3 >> $O/c4_split$sp.json 2>> $O/err.txt
done
python - <<PY
import json, glob, os
for f in sorted(glob.glob("$O/c4_*.json")):
    for l in open(f):
        if l.startswith("{"):
            j = json.loads(l); pc = j["config"]["packed_collection"]
            print(os.path.basename(f), "ms/step", j["ms_per_step"], "kernel", j["roofline"].get("avg_launch_ms"), "frac", j["roofline"].get("frac"), "bytes", pc["bytes"], "build_ms", pc["build_ms"], "count", j["config"]["result_count"], j.get("cpu_baseline", {}).get("matches_gpu_sample"))
PY
tail -3 $O/err.txt
