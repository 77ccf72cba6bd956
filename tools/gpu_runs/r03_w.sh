#!/bin/bash
# select through the select directory over the rank lines (k_select_sdir) vs the block index + octant directory (k_select_lines)
export TMPDIR=/tmp
O=gpurun_out/${1:-r03w}; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_group.py -x -q -m gpu -k "rank or select or rs_" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
for sl in 1 2; do
  BMX_RS_SELECT_LINES=$sl timeout 300 python bench.py --config 3 --no-cpu --steps 10 --warmup 3 > $O/c3_sel$sl.json 2>> $O/err.txt
done
for sh in 8 10 11; do
  BMX_RS_SDIR_SHIFT=$sh timeout 300 python bench.py --config 3 --no-cpu --steps 10 --warmup 3 > $O/c3_sel2_shift$sh.json 2>> $O/err.txt
done
for sh in 6 7 8; do BMX_RS_SDIR_SHIFT=$sh timeout 300 python bench.py --config 3 --density-q16 655 --no-cpu --steps 10 --warmup 3 > $O/c3_1pct_sel2_shift$sh.json 2>> $O/err.txt; done
timeout 300 python bench.py --config 3 --density-q16 655 --no-cpu --steps 10 --warmup 3 > $O/c3_1pct_sel2.json 2>> $O/err.txt
BMX_RS_SELECT_LINES=1 timeout 300 python bench.py --config 3 --density-q16 655 --no-cpu --steps 10 --warmup 3 > $O/c3_1pct_sel1.json 2>> $O/err.txt
python - <<PY
import json, glob, os
for f in sorted(glob.glob("$O/c3_*.json")):
    try:
        j = json.loads([l for l in open(f) if l.startswith("{")][-1])
        c = j["config"]
        print(os.path.basename(f), "rank_ms", c.get("rank_ms"), "select_ms", c.get("select_ms"), "roundtrip", c.get("rank_select_roundtrip_ok"), "frac", j["roofline"]["frac"], "select_frac", j["roofline"].get("select_frac"))
    except Exception as e: print(os.path.basename(f), "FAILED", e)
PY
tail -3 $O/err.txt
