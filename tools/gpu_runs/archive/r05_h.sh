#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r05_h; rm -rf $O; mkdir -p $O
cd /tmp; rm -rf /tmp/ks
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/ks -o s -f csv -- python $GRAFT_REPO_ROOT/bench.py --config 4 --no-cpu --no-subset --steps 3 --warmup 1 > $GRAFT_REPO_ROOT/$O/bench_under_rocprof.json 2>> $GRAFT_REPO_ROOT/$O/err.txt
cd $GRAFT_REPO_ROOT
cp $(find /tmp/ks -name "*kernel_stats.csv" | head -1) $O/kernel_stats_config4_tile_build.csv
head -12 $O/kernel_stats_config4_tile_build.csv | cut -c1-200
