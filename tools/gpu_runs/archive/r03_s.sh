#!/bin/bash
# persistent pairwise count kernel for mixed block kinds (k_count_op2_loop) vs a wave per column
export TMPDIR=/tmp
O=gpurun_out/${1:-r03s}; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pairwise" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
for pl in 4; do for t in 0 1; do
  BMX_PAIR_LOOP=$pl BMX_PAIR_NT=$t timeout 300 python bench.py --config 1 --density-q16 655 --no-cpu --steps 20 --warmup 3 > $O/c1_1pct_pl${pl}_nt$t.json 2>> $O/err.txt
  BMX_PAIR_STREAM=0 BMX_PAIR_LOOP=$pl BMX_PAIR_NT=$t timeout 300 python bench.py --config 1 --no-cpu --steps 20 --warmup 3 > $O/c1_10pct_loop_pl${pl}_nt$t.json 2>> $O/err.txt
done; done
BMX_PAIR_STREAM=0 BMX_PAIR_LOOP=0 timeout 300 python bench.py --config 1 --no-cpu --steps 20 --warmup 3 > $O/c1_10pct_percolumn.json 2>> $O/err.txt
BMX_PAIR_LOOP=4 timeout 300 python bench.py --config 1 --density-q16 200 --no-cpu --steps 20 --warmup 3 > $O/c1_03pct_pl4.json 2>> $O/err.txt
BMX_PAIR_LOOP=4 timeout 300 python bench.py --config 1 --density-q16 1311 --no-cpu --steps 20 --warmup 3 > $O/c1_2pct_pl4.json 2>> $O/err.txt
BMX_PAIR_LOOP=0 timeout 300 python bench.py --config 1 --density-q16 1311 --no-cpu --steps 20 --warmup 3 > $O/c1_2pct_pl0.json 2>> $O/err.txt
python - <<PY
import json, glob, os
for f in sorted(glob.glob("$O/c1_*.json")):
    try:
        j = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(os.path.basename(f), "frac", j["roofline"]["frac"], {k: v["kernel_ms"] for k, v in j["config"]["per_op"].items()}, j["config"]["block_types_vec0"])
    except Exception as e: print(os.path.basename(f), "FAILED", e)
PY
tail -3 $O/err.txt
