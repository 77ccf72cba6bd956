#!/bin/bash
# round 5: spill-free shapes A/B (materialised combine_and, eq-counts), tests touched so far, then the default bench line
export TMPDIR=/tmp
O=gpurun_out/r05_c; rm -rf $O; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x -k "launch_shape or and_rows or counting_formulation or batched_equality or full_size_256way or scanner_vs_numpy or uploaded_only" > $O/pytest_sel.txt 2>&1; echo "pytest rc=$?" >> $O/summary.txt
tail -3 $O/pytest_sel.txt >> $O/summary.txt
python - > $O/agg_shape.txt 2>&1 <<'PY'
import time, json, numpy as np
import bitmagic_amd as bm
ctx = bm.context(0)
vecs = [bm.bvector.generate(ctx, 0xB17A61C, v, 6554, 1_000_000_000, with_common=True) for v in range(256)]
agg = bm.aggregator(ctx)
for rnd in range(2):
    for shape in (0, 1):
        ctx.set_tuning("agg_shape", shape)
        ts = []
        for _ in range(6):
            ctx.synchronize(); t0 = time.perf_counter(); t, _ = agg.combine_and_sub(vecs, []); ts.append((time.perf_counter() - t0) * 1e3); c = t.count(); del t
        print(json.dumps({"agg_shape": shape, "host_call_ms_best": round(min(ts[1:]), 4), "median": round(float(np.median(ts[1:])), 4), "count": c}))
PY
cat $O/agg_shape.txt >> $O/summary.txt
timeout 600 python tools/bench_scanner.py > $O/bench_scanner.log 2>> $O/err.txt; tail -12 $O/bench_scanner.log >> $O/summary.txt
( time timeout 1200 python bench.py > $O/bench.json 2> $O/bench.err ) 2>> $O/summary.txt
python - <<PY >> $O/summary.txt
import json
r = json.load(open("$O/bench.json"))
print("headline", r["value"], r["ms_per_step"], r["roofline"]["frac"], r.get("cpu_baseline", {}).get("matches_gpu_full"))
print("materialised", r.get("materialised_combine_and")); print("h2d", r.get("h2d_upload"))
for k, v in r.get("other_configs", {}).items():
    print(k, json.dumps(v)[:700])
PY
