export TMPDIR=/tmp
O=gpurun_out/r03al; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "batched_equality or scanner" > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
timeout 600 python tools/bench_scanner.py > $O/bench_scanner.log 2>> $O/err.txt
python - <<PY
import json
for l in open("$O/bench_scanner.log"):
    if "scanner_transposed" in l:
        d = json.loads(l); print(d.get("queries"), d.get("table"), d.get("passes_over_the_planes"), d.get("host_call_ms"), d.get("counts_equal_pipeline"))
PY
tail -2 $O/err.txt
