#!/bin/bash
# round 4, run ak: bmx_op2_dev over operands with GAP blocks
export TMPDIR=/tmp
O=gpurun_out/${1:-r04ak}; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -q -m gpu -x -k "async_pairwise or pairwise_materialised" > $O/pytest_sel.txt 2>&1; echo "rc $?" >> $O/pytest_sel.txt; tail -15 $O/pytest_sel.txt
