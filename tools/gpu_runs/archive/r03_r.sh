#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/${1:-r03r}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "batched_equality" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
timeout 600 python tools/bench_scanner.py > $O/bench_scanner.log 2>> $O/err.txt
grep scanner_transposed $O/bench_scanner.log
cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_c1 -o c1 -- python $GRAFT_REPO_ROOT/bench.py --config 1 --no-cpu --steps 20 --warmup 3 > $GRAFT_REPO_ROOT/$O/c1_under_rocprof.json 2>> $GRAFT_REPO_ROOT/$O/err.txt
cd $GRAFT_REPO_ROOT
f=$(find $O/prof_c1 -name "*kernel_stats.csv" | head -1); head -12 $f
find $O/prof_c1 -name "*.csv" ! -name "*kernel_stats*" -delete; find $O/prof_c1 -name "*.db" -delete
