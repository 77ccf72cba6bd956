#!/bin/bash
# round 3: lean table for > 2,048 batched equality counts; streaming materialised pairwise kernel
export TMPDIR=/tmp
O=gpurun_out/${1:-r03q}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "batched_equality or materialised_streaming or pairwise or scanner" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
for w in 1 2 3 4; do
  BMX_OP2_WGS=$w timeout 300 python bench.py --config 1 --no-cpu --steps 20 --warmup 3 > $O/c1_wgs$w.json 2>> $O/err.txt
done
BMX_PAIR_STREAM=0 timeout 300 python bench.py --config 1 --no-cpu --steps 20 --warmup 3 > $O/c1_nostream.json 2>> $O/err.txt
timeout 600 python tools/bench_scanner.py > $O/bench_scanner.log 2>> $O/err.txt
python - <<PY
import json, glob, os
for f in sorted(glob.glob("$O/c1_*.json")):
    try:
        j = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(os.path.basename(f), {k: (v["kernel_ms"], v["materialised_host_call_ms"]) for k, v in j["config"]["per_op"].items()})
    except Exception as e: print(os.path.basename(f), "FAILED", e)
PY
grep scanner_transposed $O/bench_scanner.log
tail -3 $O/err.txt
