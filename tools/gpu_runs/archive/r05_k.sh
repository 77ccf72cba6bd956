#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r05_k; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x -k "in_flight_forms or full_size_rank_select or golden_case or rank_line or past_2_32" > $O/pytest_sel.txt 2>&1; echo "pytest rc=$?" >> $O/summary.txt
tail -5 $O/pytest_sel.txt >> $O/summary.txt
for lanes in 0 2; do BMX_RS_LANES=$lanes timeout 600 python bench.py --config 3 --no-cpu > $O/bench_config3_lanes$lanes.json 2>> $O/err.txt; done
timeout 600 python bench.py --config 3 --no-cpu --density-q16 655 > $O/bench_config3_1pct.json 2>> $O/err.txt
python - <<PY >> $O/summary.txt
import json
for f in ("bench_config3_lanes0", "bench_config3_lanes2", "bench_config3_1pct"):
    r = json.load(open("$O/%s.json" % f)); c = r["config"]
    print(f, "rank", c["rank_ms"], "select", c["select_ms"], "sorted", c["select_ms_sorted_ranks"], "global-dir kernel", c["select_ms_global_directory_kernel"], c["select_ms_by_batch"], "ok", c["rank_select_roundtrip_ok"], "sel frac", r["roofline"]["select"]["frac"])
PY
