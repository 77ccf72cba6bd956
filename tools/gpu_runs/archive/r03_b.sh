#!/bin/bash
# round 3, pass B: packed GAP collections + rank/select queries-in-flight forms: parity tests, then the numbers
export TMPDIR=/tmp
O=gpurun_out/${1:-r03b}; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "packed or rank_select_queries or many_gap or full_size_rank" > $O/pytest_new.txt 2>&1; echo "pytest new rc=$?" >> $O/summary.txt
tail -5 $O/pytest_new.txt >> $O/summary.txt
timeout 600 python -m pytest tests/test_gpu_group.py -x -q -m gpu > $O/pytest_group.txt 2>&1; echo "pytest group rc=$?" >> $O/summary.txt
# configs[4]: table kernel vs packed collection
for gp in 0 1; do
  BMX_GAP_PACK=$gp timeout 600 python bench.py --config 4 --steps 10 --warmup 3 --no-cpu > $O/bench_config4_pack$gp.json 2>> $O/err.txt; echo "config4 pack=$gp rc=$?" >> $O/summary.txt
done
# all-GAP 256-way AND+COUNT (0.3 % and 0.1 %): gapcount kernel vs packed collection
for dq in 197 66; do
  for gp in 0 1; do
    BMX_GAP_PACK=$gp timeout 600 python bench.py --density-q16 $dq --no-cpu --no-others --no-shard-probe --steps 10 --warmup 3 > $O/bench_dq${dq}_pack$gp.json 2>> $O/err.txt; echo "dq$dq pack=$gp rc=$?" >> $O/summary.txt
  done
done
# rank / select: queries in flight per group
for u in 1 2 4; do
  BMX_RS_UNROLL=$u timeout 600 python bench.py --config 3 --no-cpu --steps 10 > $O/bench_config3_u$u.json 2>> $O/err.txt
  BMX_RS_UNROLL=$u timeout 600 python bench.py --config 3 --no-cpu --steps 10 --density-q16 655 > $O/bench_config3_1pct_u$u.json 2>> $O/err.txt
done
python - <<'PY' >> $O/summary.txt
import json, glob, os
O = os.environ.get("O", "gpurun_out/r03b")
for f in sorted(glob.glob(O + "/bench_*.json")):
    try:
        j = json.loads([l for l in open(f) if l.startswith("{")][-1])
        c = j["config"]
        extra = {k: c.get(k) for k in ("rank_ms", "select_ms", "result_count", "build_seconds") if k in c}
        print(os.path.basename(f), "ms/step", j["ms_per_step"], "frac", j["roofline"].get("frac"), "kernel", str(j["roofline"].get("kernel"))[:60], extra)
    except Exception as e:
        print(os.path.basename(f), "FAILED", e)
PY
cat $O/summary.txt
