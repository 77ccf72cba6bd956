#!/bin/bash
# rocprofv3 --kernel-trace --stats of the round-2 secondary benches (range search, small collections, configs 1/3/4)
# usage: bash tools/gpu_runs/profile_r02i.sh <tag>      -> gpurun_out/<tag>/kernel_stats_*.csv
TAG=${1:-r02i}
export TMPDIR=/tmp
O=gpurun_out/$TAG; mkdir -p $O
prof() {   # name, command...
  local name=$1; shift
  rm -rf /tmp/prof_$name
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o $name -f csv -- "$@" > $O/$name.log 2> $O/$name.err
  find /tmp/prof_$name -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_$name.csv \;
  head -6 $O/kernel_stats_$name.csv
}
prof scanner python tools/bench_scanner.py
prof small python tools/bench_small.py
prof config1 python bench.py --config 1 --no-cpu
prof config3 python bench.py --config 3 --no-cpu
prof config4 python bench.py --config 4 --no-cpu
