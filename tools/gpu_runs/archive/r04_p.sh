#!/bin/bash
# round 4, run p: select against lanes per query x select-directory granularity (profiles/r04_select)
export TMPDIR=/tmp
O=gpurun_out/${1:-r04p}; mkdir -p $O
for lanes in 2 4; do for sh in 0 11 12; do
  BMX_RS_LANES=$lanes BMX_RS_SDIR_SHIFT=$sh timeout 600 python bench.py --config 3 --no-cpu --steps 5 --warmup 2 > $O/c3.json 2>> $O/err.txt
  python - <<PY
import json
j = json.loads([l for l in open("$O/c3.json") if l.startswith("{")][-1])
c = j["config"]; print("lanes $lanes shift $sh", "rank", c["rank_ms"], "select", c["select_ms"], "ok", c["rank_select_roundtrip_ok"])
PY
done; done
