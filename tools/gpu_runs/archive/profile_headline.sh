#!/bin/bash
# rocprofv3 evidence for the headline bench: --stats pass + separate PMC passes (FETCH_SIZE, WRITE_SIZE, TCC, SQ)
# usage: bash tools/gpu_runs/profile_headline.sh <tag>      -> gpurun_out/<tag>/
set -x
TAG=${1:-r02c}
export TMPDIR=/tmp
O=gpurun_out/$TAG; rm -rf $O; mkdir -p $O
timeout 900 rocprofv3 --kernel-trace --stats -d $O/stats -o $TAG -f csv -- python bench.py --no-cpu --no-shard-probe > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o $TAG -f csv -- python bench.py --steps 3 --warmup 1 --no-cpu --no-shard-probe > /dev/null 2> $O/pmc_fetch.err
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o $TAG -f csv -- python bench.py --steps 3 --warmup 1 --no-cpu --no-shard-probe > /dev/null 2> $O/pmc_write.err
timeout 900 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d $O/pmc_tcc -o $TAG -f csv -- python bench.py --steps 3 --warmup 1 --no-cpu --no-shard-probe > /dev/null 2> $O/pmc_tcc.err
timeout 900 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $O/pmc_sq -o $TAG -f csv -- python bench.py --steps 3 --warmup 1 --no-cpu --no-shard-probe > /dev/null 2> $O/pmc_sq.err
for d in stats pmc_fetch pmc_write pmc_tcc pmc_sq; do
  for f in $(find $O/$d -name "*kernel_trace.csv" -o -name "*counter_collection.csv"); do
    (head -1 $f; grep k_pipe_counts $f) > $O/${d}_$(basename $f .csv)_k_pipe_counts.csv; rm $f
  done
done
find $O -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_bench.csv \;
head -5 $O/kernel_stats_bench.csv
python bench.py --no-cpu > $O/bench_unprofiled.json 2> /dev/null
cat $O/bench_unprofiled.json
