#!/bin/bash
# select through the select directory: line loads non-temporal (directory stays in the L2) vs plain; coarser / finer directories
export TMPDIR=/tmp
O=gpurun_out/${1:-r03w}; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_group.py -x -q -m gpu -k "rank or select or rs_" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
for nt in 1 0; do
  BMX_RS_NT=$nt timeout 300 python bench.py --config 3 --no-cpu --steps 10 --warmup 3 > $O/c3_nt$nt.json 2>> $O/err.txt
  BMX_RS_NT=$nt timeout 300 python bench.py --config 3 --density-q16 655 --no-cpu --steps 10 --warmup 3 > $O/c3_1pct_nt$nt.json 2>> $O/err.txt
done
for sh in 8 9 11; do BMX_RS_NT=1 BMX_RS_SDIR_SHIFT=$sh timeout 300 python bench.py --config 3 --no-cpu --steps 10 --warmup 3 > $O/c3_nt1_shift$sh.json 2>> $O/err.txt; done
python - <<PY
import json, glob, os
for f in sorted(glob.glob("$O/c3_*.json")):
    try:
        j = json.loads([l for l in open(f) if l.startswith("{")][-1])
        c = j["config"]
        print(os.path.basename(f), "rank_ms", c.get("rank_ms"), "select_ms", c.get("select_ms"), "roundtrip", c.get("rank_select_roundtrip_ok"), "frac", j["roofline"]["frac"], "select_frac", j["roofline"]["select"]["frac"])
    except Exception as e: print(os.path.basename(f), "FAILED", e)
PY
tail -3 $O/err.txt
