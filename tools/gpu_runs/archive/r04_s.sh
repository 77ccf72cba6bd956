#!/bin/bash
# round 4, run s: the row kernel over slotted slabs: plain / non-temporal rows, 4 / 8 in flight, and its load side alone
export TMPDIR=/tmp
O=gpurun_out/${1:-r04s}; mkdir -p $O
run() { name=$1; shift
  env "$@" timeout 600 python bench.py --config 4 --no-cpu --steps 6 --warmup 2 > $O/c4_$name.json 2> $O/c4_$name.err
  python - <<PY
import json
try:
    j = json.loads([l for l in open("$O/c4_$name.json") if l.startswith("{")][-1])
    print("$name", j["ms_per_step"], j["config"]["cold_ms"], j["config"]["result_count"])
except Exception as e:
    print("$name", "failed", e); print(open("$O/c4_$name.err").read()[-600:])
PY
}
run plain_d4 BMX_OR_NT=0 BMX_OR_DEPTH=4
run nt_d4 BMX_OR_NT=1 BMX_OR_DEPTH=4
run nt_d8 BMX_OR_NT=1 BMX_OR_DEPTH=8
run tiled BMX_OR_ROWS=0
export BMX_LIB=bitmagic_amd/lib/libbmx_tune.so
run loads_plain_d4 BMX_OR_NT=0 BMX_DIAG_ROWS=512
run loads_nt_d4 BMX_OR_NT=1 BMX_DIAG_ROWS=512
run loads_nt_d8 BMX_OR_NT=1 BMX_OR_DEPTH=8 BMX_DIAG_ROWS=512
