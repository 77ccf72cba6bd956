#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r05_q; rm -rf $O; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "gap_results_converted" ) > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/summary.txt
tail -15 $O/pytest.txt >> $O/summary.txt
cat $O/summary.txt
