#!/bin/bash
# round 4, run ad: result_finish without its trailing synchronise: whole GPU suite, both soaks, small-collection and pairwise timings
export TMPDIR=/tmp
O=gpurun_out/${1:-r04ad}; rm -rf $O; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc $?" | tee -a $O/summary.txt
grep -E "passed|failed" $O/pytest.txt | tee -a $O/summary.txt
timeout 900 python tools/soak_r04.py 120 > $O/soak_r04.log 2>&1; tail -1 $O/soak_r04.log | tee -a $O/summary.txt
bash tools/gpu_runs/soak.sh > /dev/null 2>&1; tail -2 gpurun_out/soak.log | tee -a $O/summary.txt; cp gpurun_out/soak.log $O/soak.log
timeout 600 python tools/op2_ab.py 655 2>/dev/null | tee -a $O/summary.txt
timeout 600 python tools/bench_small.py > $O/bench_small.log 2>> $O/err.txt; grep -i "combine_and\|combine_or\|find_first" $O/bench_small.log | head -12
