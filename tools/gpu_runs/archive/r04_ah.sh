#!/bin/bash
# round 4, run ah: k_emit_gaps_list (GAP candidates named by a list): pairwise tests, soak, per-op host-call times
export TMPDIR=/tmp
O=gpurun_out/${1:-r04ah}; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -q -m gpu -x -k "pairwise or op2 or stress or golden or random_block or async" > $O/pytest_sel.txt 2>&1; echo "rc $?" >> $O/pytest_sel.txt; tail -3 $O/pytest_sel.txt
timeout 600 python tools/soak_r04.py 45 2>&1 | tail -2
for i in 1 2; do timeout 600 python tools/op2_ab.py 655 2>/dev/null | tee -a $O/op2.jsonl; done
