#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r05_j; rm -rf $O; mkdir -p $O
cd /tmp; rm -rf /tmp/ks
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/ks -o s -f csv -- python $GRAFT_REPO_ROOT/tools/prof_prepare_or.py > $GRAFT_REPO_ROOT/$O/out.txt 2>&1
cd $GRAFT_REPO_ROOT
cp $(find /tmp/ks -name "*kernel_stats.csv" | head -1) $O/kernel_stats_prepare_or.csv
grep -E "k_coll|k_build_tdir" $O/kernel_stats_prepare_or.csv | sed "s/(.*)\"//" | cut -c1-120; grep build_ms $O/out.txt
