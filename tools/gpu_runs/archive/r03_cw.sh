#!/bin/bash
# packed collections: a wave's batch as ONE 4-KiB piece (coll_shape 4, 5) vs four 1-KiB pieces 8 KiB apart (2, 3)
export TMPDIR=/tmp
O=gpurun_out/${1:-r03ag}; mkdir -p $O
for sh in 4 5; do BMX_COLL_SHAPE=$sh timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "packed_gap_collections" > $O/pytest_sh$sh.txt 2>&1; tail -1 $O/pytest_sh$sh.txt; done
for sh in 2 4 5 2 4; do
  BMX_COLL_SHAPE=$sh timeout 600 python bench.py --config 4 --steps 10 --warmup 3 --no-cpu >> $O/c4_shape$sh.json 2>> $O/err.txt
  BMX_COLL_SHAPE=$sh timeout 600 python bench.py --density-q16 197 --no-cpu --no-others --no-shard-probe --steps 10 --warmup 3 >> $O/dq197_shape$sh.json 2>> $O/err.txt
done
python - <<PY
import json, glob, os
for f in sorted(glob.glob("$O/*.json")):
    for l in open(f):
        if l.startswith("{"):
            j = json.loads(l); print(os.path.basename(f), "ms/step", j["ms_per_step"], "kernel", j["roofline"].get("avg_launch_ms"), "frac", j["roofline"].get("frac"))
PY
tail -3 $O/err.txt
