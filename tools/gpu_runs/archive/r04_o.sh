#!/bin/bash
# round 4, run o: select directory granularity (ones per entry = 2^shift) on configs[3]
export TMPDIR=/tmp
O=gpurun_out/${1:-r04o}; mkdir -p $O
for sh in 0 11 12 13 14 15; do
  BMX_RS_SDIR_SHIFT=$sh timeout 600 python bench.py --config 3 --no-cpu --steps 5 --warmup 2 > $O/c3_sh$sh.json 2>> $O/err.txt
  python - <<PY
import json
j = json.loads([l for l in open("$O/c3_sh$sh.json") if l.startswith("{")][-1])
c = j["config"]; print("shift $sh", "rank", c["rank_ms"], "select", c["select_ms"], "ok", c["rank_select_roundtrip_ok"], "bytes", c["hbm_resident_bytes"])
PY
done
tail -2 $O/err.txt
