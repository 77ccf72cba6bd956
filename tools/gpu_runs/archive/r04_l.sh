#!/bin/bash
# round 4, run l: materialised pairwise ops over mixed block kinds: parity + configs[1] at 1 % (persistent kernel vs a wave per column)
export TMPDIR=/tmp
O=gpurun_out/${1:-r04l}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pairwise" > $O/pytest.log 2>&1; tail -4 $O/pytest.log
for cfg in "loop:BMX_OP2_LOOP=-1" "loop8:BMX_OP2_LOOP=8" "loop2:BMX_OP2_LOOP=2" "percol:BMX_OP2_LOOP=0"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  env $envs timeout 600 python bench.py --config 1 --density-q16 655 --no-cpu --steps 5 --warmup 2 > $O/c1_1pct_$name.json 2>> $O/err.txt
  python - <<PY
import json
j = json.loads([l for l in open("$O/c1_1pct_$name.json") if l.startswith("{")][-1])
print("$name", {k: (v["materialised_host_call_ms"], v["materialised_GBps"]) for k, v in j["config"]["per_op"].items()})
PY
done
tail -2 $O/err.txt
