#!/bin/bash
# round 4, run c: the row kernel (k_agg_or_rows): parity, then configs[4] cold against the column-tile kernel
export TMPDIR=/tmp
O=gpurun_out/${1:-r04c}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "row_kernel or many_gap_operands or packed_gap" > $O/pytest.log 2>&1
tail -5 $O/pytest.log
for cfg in "rows1_d4:BMX_OR_ROWS=1 BMX_OR_DEPTH=4" "rows1_d8:BMX_OR_ROWS=1 BMX_OR_DEPTH=8" "rows1_d4_noswz:BMX_OR_ROWS=1 BMX_OR_DEPTH=4 BMX_XCD_SWIZZLE=0" "rows0:BMX_OR_ROWS=0"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  env $envs timeout 600 python bench.py --config 4 --no-cpu --steps 6 --warmup 2 > $O/c4_$name.json 2> $O/c4_$name.err
  python - <<PY
import json
try:
    j = json.loads([l for l in open("$O/c4_$name.json") if l.startswith("{")][-1])
    print("$name", j["ms_per_step"], j["roofline"]["avg_launch_ms"], j["config"]["result_count"])
except Exception as e:
    print("$name", "failed", e); print(open("$O/c4_$name.err").read()[-800:])
PY
done
