#!/bin/bash
# round 4, run n: dense materialised pairwise ops (configs[1] at 10 %) next to the library's own 2R:1W probe
export TMPDIR=/tmp
O=gpurun_out/${1:-r04n}; mkdir -p $O
for w in 2 4 8; do
  BMX_OP2_WGS=$w timeout 600 python bench.py --config 1 --no-cpu --steps 5 --warmup 2 > $O/c1_w$w.json 2>> $O/err.txt
  python - <<PY
import json
j = json.loads([l for l in open("$O/c1_w$w.json") if l.startswith("{")][-1])
print("wgs $w", {k: v["materialised_host_call_ms"] for k, v in j["config"]["per_op"].items()}, j["config"]["own_read_write_probe"], (j["config"]["read_write_probe"] or {}).get("ms"))
PY
done
timeout 600 rocprofv3 --kernel-trace --stats -d $O/p -o x -f csv -- python bench.py --config 1 --no-cpu --steps 5 --warmup 2 > /dev/null 2>> $O/err.txt
f=$(find $O/p -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats_config1.csv; grep -E "k_op2_stream|k_probe_rw|vectorized_elementwise" $O/kernel_stats_config1.csv | sed 's/(.*)"/"/' | cut -c1-150; rm -rf $O/p
tail -2 $O/err.txt
