#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r05_g; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x -k "tile_build or packed or prepared or full_size_or" > $O/pytest_sel.txt 2>&1; echo "pytest rc=$?" >> $O/summary.txt
tail -12 $O/pytest_sel.txt >> $O/summary.txt
for b in 1 0; do BMX_COLL_BUILD=$b timeout 600 python bench.py --config 4 --no-cpu --no-subset > $O/bench_config4_build$b.json 2>> $O/err.txt; done
python - <<PY >> $O/summary.txt
import json
for b in (1, 0):
    r = json.load(open("$O/bench_config4_build%d.json" % b)); c = r["config"]
    print("coll_build", b, "cold", c["cold_ms"], "build", c["build_ms"], "prepare wall", c["prepare_call_wall_ms"], "warm", c["warm_ms"], "break-even", c["break_even_calls"], "warm==cold", c["warm_count_equal"], c["packed_collection"])
PY
