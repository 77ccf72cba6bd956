#!/bin/bash
# round 4, run ai: persistent form of k_agg_or_rows: parity, then A/B against the workgroup-per-tile form on one box
export TMPDIR=/tmp
O=gpurun_out/${1:-r04ai}; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -q -m gpu -x -k "row_kernel or full_size_or" > $O/pytest_sel.txt 2>&1; echo "rc $?" >> $O/pytest_sel.txt; tail -4 $O/pytest_sel.txt
timeout 300 python tools/soak_r04.py 40 2>&1 | grep "A done"
for rep in 1 2; do for p in 1 0; do
  BMX_OR_PERSIST=$p timeout 300 python tools/tail_probe.py 4360 2>/dev/null | sed "s/^/persist=$p /" | tee -a $O/ab.txt
done; done
