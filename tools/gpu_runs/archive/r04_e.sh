#!/bin/bash
# round 4, run e: the row kernel's load side alone (tuning build, BMX_DIAG_ROWS=512) next to the full kernel
export TMPDIR=/tmp
O=gpurun_out/${1:-r04e}; mkdir -p $O
export BMX_LIB=bitmagic_amd/lib/libbmx_tune.so BMX_GAP_PACK=0 BMX_OR_ROWS=1
for cfg in "full_d4:BMX_OR_DEPTH=4" "loads_d4:BMX_OR_DEPTH=4 BMX_DIAG_ROWS=512" "loads_d8:BMX_OR_DEPTH=8 BMX_DIAG_ROWS=512"; do
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
