#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r05_d; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_group.py -q -x -k "limit or golden_case or launch_shape or scanner or pipeline" > $O/pytest_sel.txt 2>&1; echo "pytest rc=$?" >> $O/summary.txt
tail -15 $O/pytest_sel.txt >> $O/summary.txt
