#!/bin/bash
# round 6: everything DESIGN.md's current-numbers table quotes, in ONE pass on one box with the library as committed: GPU tests, smoke,
# the default bench line (headline + CPU legs + the other configs' summaries) exactly as the driver runs it, every config on its own,
# the sparse first-call variants, small collections / scanner / shift / select table, and the rocprofv3 evidence: --kernel-trace --stats
# of the bench commands and the PMC passes (FETCH_SIZE, WRITE_SIZE, TCC hit / miss in separate passes) that tools/make_traffic_json.py
# turns into profiles/traffic_*.json.  summary.txt is written LAST, from the files of this pass only.
export TMPDIR=/tmp
O=gpurun_out/${1:-r06final}; rm -rf $O; mkdir -p $O
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
stats_of() {  # stats_of <name> <command...>
  local name=$1; shift
  rm -rf /tmp/ks; ( cd /tmp; timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/ks -o s -f csv -- "$@" > $R/$O/${name}_under_rocprof.txt 2>> $R/$O/err.txt )
  cp $(find /tmp/ks -name "*kernel_stats.csv" | head -1) $O/kernel_stats_$name.csv
  # the same trace split by launch shape: a kernel that the run calls at several sizes (full-size, subset, warm-up) gets one line per grid
  python - "$(find /tmp/ks -name '*kernel_trace.csv' | head -1)" > $O/kernel_stats_${name}_by_shape.csv <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
try:
    for r in csv.DictReader(open(sys.argv[1])):
        g = tuple(int(r.get(k, 0) or 0) for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z"))
        acc[(r["Kernel_Name"][:60], g[0] * max(g[1], 1) * max(g[2], 1))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
except Exception as e:
    print("kernel trace not split:", e)
print("kernel,grid_threads,calls,avg_ms,min_ms,max_ms")
for (k, g), v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    print('"%s",%d,%d,%.6f,%.6f,%.6f' % (k, g, len(v), sum(v) / len(v), min(v), max(v)))
PY
}
B="python $R/bench.py"
( time timeout 2400 python -m pytest tests -q -m gpu > $O/pytest.txt 2>&1 ) 2> $O/pytest_time.txt; echo "pytest gpu rc=$?" > $O/rc.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?" >> $O/rc.txt
( time timeout 1200 python bench.py > $O/bench.json 2> $O/bench.err ) 2> $O/bench_time.txt
for c in 0 1 3 4; do timeout 600 python bench.py --config $c > $O/bench_config$c.json 2>> $O/err.txt; done
timeout 600 python bench.py --config 1 --density-q16 655 > $O/bench_config1_1pct.json 2>> $O/err.txt
timeout 600 python bench.py --config 1 --density-q16 32768 > $O/bench_config1_50pct.json 2>> $O/err.txt
timeout 600 python bench.py --config 3 --density-q16 655 > $O/bench_config3_1pct.json 2>> $O/err.txt
for dq in 328 197 66; do timeout 600 python bench.py --density-q16 $dq --no-prepare --no-cpu --no-others --no-shard-probe > $O/bench_dq${dq}_first_call.json 2>> $O/err.txt; done
for dq in 197 66; do timeout 600 python bench.py --density-q16 $dq --no-cpu --no-others --no-shard-probe > $O/bench_dq${dq}_prepared.json 2>> $O/err.txt; done
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu --no-others > $O/bench_torchrun1.json 2> $O/bench_torchrun1.err
BMX_BENCH_TEST_ONE_DEVICE=1 timeout 600 python bench.py --gpus 8 --nvec 64 --steps 5 --warmup 2 --no-cpu > $O/bench_group8_hook.json 2>> $O/err.txt
timeout 600 python tools/bench_small.py > $O/bench_small.log 2>> $O/err.txt
timeout 600 python tools/bench_scanner.py > $O/bench_scanner.log 2>> $O/err.txt
timeout 600 python tools/bench_shift.py > $O/bench_shift.log 2>> $O/err.txt
timeout 600 python tools/bench_pipeline_coll.py > $O/bench_pipeline_coll.log 2>> $O/err.txt
timeout 600 python tools/bench_select6.py > $O/select_table.jsonl 2>> $O/err.txt
timeout 600 python tools/bench_and_rows.py > $O/and_rows_shapes.jsonl 2>> $O/err.txt
timeout 900 python tools/soak_r05.py 200 > $O/soak_r05.txt 2>&1
( export BMX_DEBUG_REDZONE=1; timeout 1200 python -m pytest tests -q -m gpu > $O/pytest_redzone.txt 2>&1; timeout 900 python tools/soak_r05.py 120 > $O/soak_r05_redzone.txt 2>&1; timeout 900 python tools/soak_r04.py 40 > $O/soak_r04_redzone.txt 2>&1 )
timeout 900 python tools/soak_r04.py 40 > $O/soak_r04.txt 2>&1
timeout 900 python tools/soak_r06.py 60 > $O/soak_r06.txt 2>&1
# rocprofv3 --stats of the bench commands
stats_of bench $B --no-cpu --no-others --no-shard-probe
stats_of config1 $B --config 1 --no-cpu
stats_of config1_1pct $B --config 1 --density-q16 655 --no-cpu
stats_of config3 $B --config 3 --no-cpu
stats_of config4 $B --config 4 --no-cpu
stats_of dq197_first_call $B --density-q16 197 --no-prepare --no-cpu --no-others --no-shard-probe
stats_of dq66_first_call $B --density-q16 66 --no-prepare --no-cpu --no-others --no-shard-probe
stats_of prepare_or python $R/tools/prof_prepare_or.py
# PMC passes
pmc_of $O/pmc_headline.txt k_pipe_counts_bits2 -- $B --steps 3 --warmup 1 --no-cpu --no-others --no-shard-probe
pmc_of $O/pmc_config1.txt "k_count_op2_stream|k_op2_stream" -- $B --config 1 --no-cpu --steps 3 --warmup 1
pmc_of $O/pmc_config1_1pct.txt "k_count_op2_loop|k_op2_loop" -- $B --config 1 --density-q16 655 --no-cpu --steps 3 --warmup 1
pmc_of $O/pmc_config1_50pct.txt "k_count_op2_stream|k_op2_stream" -- $B --config 1 --density-q16 32768 --no-cpu --steps 3 --warmup 1
pmc_of $O/pmc_config3.txt "k_rank|k_select|k_probe_lines" -- $B --config 3 --no-cpu --steps 3 --warmup 1
pmc_of $O/pmc_config3_1pct.txt "k_rank|k_select|k_probe_lines" -- $B --config 3 --density-q16 655 --no-cpu --steps 3 --warmup 1
pmc_of $O/pmc_config4.txt "k_agg_or_rows|k_coll_apply|k_coll2" -- $B --config 4 --no-cpu --no-subset --steps 4 --warmup 2
pmc_of $O/pmc_dq197.txt "k_agg_and_rows" -- $B --density-q16 197 --no-prepare --no-cpu --no-others --no-shard-probe --steps 3 --warmup 1
pmc_of $O/pmc_dq66.txt "k_agg_and_rows" -- $B --density-q16 66 --no-prepare --no-cpu --no-others --no-shard-probe --steps 3 --warmup 1
# summary LAST, from this pass's files only
cat $O/rc.txt > $O/summary.txt
grep -E "passed|failed" $O/pytest.txt >> $O/summary.txt; cat $O/pytest_time.txt $O/bench_time.txt | grep real >> $O/summary.txt
tail -2 $O/soak_r05.txt >> $O/summary.txt; tail -1 $O/soak_r04.txt >> $O/summary.txt; tail -2 $O/soak_r06.txt >> $O/summary.txt
echo "red zones (BMX_DEBUG_REDZONE=1): $(grep -E "passed|failed" $O/pytest_redzone.txt | tail -1); reports: $(cat $O/pytest_redzone.txt $O/soak_r05_redzone.txt $O/soak_r04_redzone.txt | grep -c "bmx redzone"); $(tail -1 $O/soak_r05_redzone.txt); $(tail -1 $O/soak_r04_redzone.txt)" >> $O/summary.txt
python - <<PY >> $O/summary.txt
import json, glob, os
for f in sorted(glob.glob("$O/bench*.json")):
    try:
        j = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(os.path.basename(f), j["value"], j["unit"], "ms/step", j["ms_per_step"], "n_gpus", j["n_gpus"], "frac", j["roofline"].get("frac"), str(j["roofline"].get("kernel"))[:46])
    except Exception as e:
        print(os.path.basename(f), "FAILED", e)
r = json.load(open("$O/bench.json"))
print("materialised_combine_and", r.get("materialised_combine_and", {}).get("host_call_ms"), "h2d", r.get("h2d_upload", {}).get("bmx_vec_upload_GBps"))
for k, v in r.get("other_configs", {}).items():
    print(" other", k, v.get("ms_per_step"), v.get("roofline", {}).get("frac"), "traffic", v.get("roofline", {}).get("traffic"), {x: v.get(x) for x in ("cold_ms", "build_ms", "warm_ms", "break_even_calls") if x in v}, v.get("cpu_baseline", {}).get("matches_gpu_full"), v.get("error"))
PY
for k in bench config1 config1_1pct config3 config4 dq197_first_call dq66_first_call prepare_or; do echo "== kernel_stats_$k" >> $O/summary.txt; head -6 $O/kernel_stats_$k.csv | cut -c1-150 >> $O/summary.txt; done
cat $O/summary.txt
