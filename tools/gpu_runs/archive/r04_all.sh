#!/bin/bash
# round 4: everything DESIGN.md's current-numbers table quotes, in one pass on one box: GPU tests, smoke, the default bench line
# (headline + CPU legs + the other configs' summaries) exactly as the driver runs it, every config on its own (with its CPU
# checks), GAP-heavy variants, entry modes, small collections / scanner / shift, and the rocprofv3 evidence: --kernel-trace
# --stats of the bench commands and the PMC passes (FETCH_SIZE, WRITE_SIZE, TCC hit / miss in separate passes) that
# tools/make_traffic_json.py turns into profiles/traffic_*.json
export TMPDIR=/tmp
O=gpurun_out/${1:-r04final}; rm -rf $O; mkdir -p $O
pmc_of() {  # pmc_of <out.txt> <kernel substrings a|b> -- <bench args...>
  local out=$1 kern=$2; shift 3
  for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    rm -rf /tmp/pmc_x
    timeout 600 rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc_x -o x -f csv -- python bench.py "$@" > /dev/null 2>> $O/pmc.err
    python - "$(find /tmp/pmc_x -name '*counter_collection.csv' | head -1)" "$kern" >> $out <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if any(k in r["Kernel_Name"] for k in sys.argv[2].split("|")): acc[(r["Kernel_Name"][:48], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()): print(k[0], k[1], "per launch avg", sum(v) / len(v), "launches", len(v))
PY
  done
}
stats_of() {  # stats_of <name> <bench args...>
  local name=$1; shift
  rm -rf /tmp/ks; timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/ks -o s -f csv -- python bench.py "$@" > $O/bench_${name}_under_rocprof.json 2>> $O/err.txt
  cp $(find /tmp/ks -name "*kernel_stats.csv" | head -1) $O/kernel_stats_$name.csv
}
( time timeout 2400 python -m pytest tests -q -m gpu > $O/pytest.txt 2>&1 ) 2>> $O/summary.txt; echo "pytest gpu rc=$?" >> $O/summary.txt
grep -E "passed|failed" $O/pytest.txt >> $O/summary.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?" >> $O/summary.txt
( time timeout 900 python bench.py > $O/bench.json 2> $O/bench.err ) 2>> $O/summary.txt
for c in 0 1 3 4; do timeout 600 python bench.py --config $c > $O/bench_config$c.json 2>> $O/err.txt; done
timeout 600 python bench.py --config 1 --density-q16 655 > $O/bench_config1_1pct.json 2>> $O/err.txt
timeout 600 python bench.py --config 1 --density-q16 32768 > $O/bench_config1_50pct.json 2>> $O/err.txt
timeout 600 python bench.py --config 3 --density-q16 655 > $O/bench_config3_1pct.json 2>> $O/err.txt
for dq in 328 197 66; do timeout 600 python bench.py --density-q16 $dq --no-cpu --no-others --no-shard-probe > $O/bench_dq$dq.json 2>> $O/err.txt; done
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu --no-others > $O/bench_torchrun1.json 2> $O/bench_torchrun1.err
BMX_BENCH_TEST_ONE_DEVICE=1 timeout 600 python bench.py --gpus 2 --nvec 64 --steps 5 --warmup 2 --no-cpu > $O/bench_group2_hook.json 2>> $O/err.txt
BMX_BENCH_TEST_ONE_DEVICE=1 timeout 600 python bench.py --gpus 8 --nvec 64 --steps 5 --warmup 2 --no-cpu > $O/bench_group8_hook.json 2>> $O/err.txt
timeout 600 python tools/bench_small.py > $O/bench_small.log 2>> $O/err.txt
timeout 600 python tools/bench_scanner.py > $O/bench_scanner.log 2>> $O/err.txt
timeout 600 python tools/bench_shift.py > $O/bench_shift.log 2>> $O/err.txt
timeout 600 python tools/bench_pipeline_coll.py > $O/bench_pipeline_coll.log 2>> $O/err.txt
# rocprofv3 --stats of the bench commands
stats_of bench --no-cpu --no-others --no-shard-probe
stats_of config1 --config 1 --no-cpu
stats_of config1_1pct --config 1 --density-q16 655 --no-cpu
stats_of config3 --config 3 --no-cpu
stats_of config4 --config 4 --no-cpu
stats_of dq197 --density-q16 197 --no-cpu --no-others --no-shard-probe
# PMC passes
pmc_of $O/pmc_headline.txt k_pipe_counts_bits2 -- --steps 3 --warmup 1 --no-cpu --no-others --no-shard-probe
pmc_of $O/pmc_config1.txt "k_count_op2_stream|k_op2_stream" -- --config 1 --no-cpu --steps 3 --warmup 1
pmc_of $O/pmc_config1_1pct.txt "k_count_op2_loop|k_op2_loop" -- --config 1 --density-q16 655 --no-cpu --steps 3 --warmup 1
pmc_of $O/pmc_config3.txt "k_rank|k_select|k_probe_lines" -- --config 3 --no-cpu --steps 3 --warmup 1
pmc_of $O/pmc_config4.txt "k_agg_or_rows|k_coll_apply" -- --config 4 --no-cpu --steps 4 --warmup 2
python - <<PY >> $O/summary.txt
import json, glob, os
for f in sorted(glob.glob("$O/bench*.json")):
    try:
        j = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(os.path.basename(f), j["value"], j["unit"], "ms/step", j["ms_per_step"], "n_gpus", j["n_gpus"], "frac", j["roofline"].get("frac"), str(j["roofline"].get("kernel"))[:46])
    except Exception as e:
        print(os.path.basename(f), "FAILED", e)
PY
cat $O/summary.txt
