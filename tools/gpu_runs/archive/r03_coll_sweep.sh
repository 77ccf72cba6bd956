#!/bin/bash
# packed-collection kernel shapes: 256 / 512 threads, prefetch, launch windows
export TMPDIR=/tmp
O=gpurun_out/${1:-r03i}; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "packed" > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
for sh in 0 1 2 3; do
  BMX_COLL_SHAPE=$sh timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "packed_gap_collections" > $O/pytest_sh$sh.txt 2>&1; tail -1 $O/pytest_sh$sh.txt
  BMX_COLL_SHAPE=$sh timeout 600 python bench.py --config 4 --steps 10 --warmup 3 --no-cpu > $O/c4_sh$sh.json 2>> $O/err.txt
  for dq in 197 66; do
    BMX_COLL_SHAPE=$sh timeout 600 python bench.py --density-q16 $dq --no-cpu --no-others --no-shard-probe --steps 10 --warmup 3 > $O/dq${dq}_sh$sh.json 2>> $O/err.txt
  done
done
for win in 2048 4096 8192; do
  BMX_COLL_WINDOW=$win timeout 600 python bench.py --config 4 --steps 10 --warmup 3 --no-cpu > $O/c4_win$win.json 2>> $O/err.txt
  BMX_COLL_WINDOW=$win timeout 600 python bench.py --density-q16 197 --no-cpu --no-others --no-shard-probe --steps 10 --warmup 3 > $O/dq197_win$win.json 2>> $O/err.txt
done
python - <<PY
import json, glob, os
for f in sorted(glob.glob("$O/*.json")):
    try:
        j = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(os.path.basename(f), "ms/step", j["ms_per_step"], "kernel_ms", j["roofline"].get("avg_launch_ms"), "frac", j["roofline"].get("frac"))
    except Exception as e: print(os.path.basename(f), "FAILED", e)
PY
