#!/bin/bash
# round 6: everything DESIGN.md's current-numbers table quotes, in ONE pass on one box with the library as committed: GPU tests, smoke,
# the default bench line (headline + CPU legs + the other configs' summaries) exactly as the driver runs it, every config on its own,
# the sparse first-call variants, small collections / scanner / shift / select table, and the rocprofv3 evidence: --kernel-trace --stats
# of the bench commands and the PMC passes (FETCH_SIZE, WRITE_SIZE, TCC hit / miss in separate passes) that tools/make_traffic_json.py
# turns into profiles/traffic_*.json.  summary.txt is written LAST, from the files of this pass only.
export TMPDIR=/tmp
O=gpurun_out/${1:-r06final}; mkdir -p $O
R=$PWD
pmc_of() {  # pmc_of <out.txt> <kernel substrings a|b> -- <command...>
  local out=$1 kern=$2; shift 3
  for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum"; do
    rm -rf /tmp/pmc_x
    ( cd /tmp; timeout 600 rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc_x -o x -f csv -- "$@" > /dev/null 2>> $R/$O/pmc.err )
    python - "$(find /tmp/pmc_x -name '*counter_collection.csv' | head -1)" "$kern" >> $out <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
try:
    for r in csv.DictReader(open(sys.argv[1])):
        if any(k in r["Kernel_Name"] for k in sys.argv[2].split("|")): acc[(r["Kernel_Name"][:48], r["Counter_Name"], int(r.get("Grid_Size", 0) or 0))].append(float(r["Counter_Value"]))
except Exception as e:
    print("pmc pass failed:", e)
# one line per (kernel, counter) over the launches of its LARGEST grid (the full-size calls: a run also launches a kernel on small
# batches / subsets, which must not dilute the per-launch figure); the other launch shapes follow as "# shape" lines
big = {}
for (k, c, g), v in acc.items():
    if (k, c) not in big or g > big[(k, c)][0]: big[(k, c)] = (g, v)
for (k, c), (g, v) in sorted(big.items()): print(k, c, "per launch avg", sum(v) / len(v), "launches", len(v))
for (k, c, g), v in sorted(acc.items()):
    if big[(k, c)][0] != g: print("# shape", k, c, "grid", g, "avg", sum(v) / len(v), "launches", len(v))
PY
  done
}
B="python $R/bench.py"
rm -f $O/pmc_config3.txt $O/pmc_config3_1pct.txt
pmc_of $O/pmc_config3.txt "k_rank|k_select|k_probe_lines" -- $B --config 3 --no-cpu --steps 3 --warmup 1
pmc_of $O/pmc_config3_1pct.txt "k_rank|k_select|k_probe_lines" -- $B --config 3 --density-q16 655 --no-cpu --steps 3 --warmup 1
cat $O/pmc_config3.txt $O/pmc_config3_1pct.txt
