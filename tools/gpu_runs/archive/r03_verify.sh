#!/bin/bash
# quick verification after small changes: rank / select / scanner tests, configs[3], the 8,192-value equality batch
export TMPDIR=/tmp
O=gpurun_out/${1:-r03ai}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_group.py -x -q -m gpu -k "rank or select or rs_ or equality or scanner or packed_collection_kernel_shapes" > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
timeout 300 python bench.py --config 3 --no-cpu --steps 10 --warmup 3 > $O/bench_config3.json 2>> $O/err.txt
timeout 600 python tools/bench_scanner.py > $O/bench_scanner.log 2>> $O/err.txt
python - <<PY
import json
j = json.loads([l for l in open("$O/bench_config3.json") if l.startswith("{")][-1]); print("config3 rank", j["config"]["rank_ms"], "select", j["config"]["select_ms"], j["config"]["rank_select_roundtrip_ok"])
for l in open("$O/bench_scanner.log"):
    if "scanner_transposed" in l:
        d = json.loads(l); print(d.get("queries"), d.get("table"), d.get("passes_over_the_planes"), d.get("host_call_ms"))
PY
tail -2 $O/err.txt
