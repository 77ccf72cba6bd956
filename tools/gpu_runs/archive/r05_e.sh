#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r05_e; rm -rf $O; mkdir -p $O
( time timeout 1500 python -m pytest tests/test_gpu_stress.py tests/test_gpu_group.py -q -x -k "soak or rccl" ) > $O/pytest_sel.txt 2>&1; echo "pytest rc=$?" >> $O/summary.txt
tail -12 $O/pytest_sel.txt >> $O/summary.txt
( time timeout 900 python tools/soak_r05.py 60 ) > $O/soak_r05.txt 2>&1; tail -4 $O/soak_r05.txt >> $O/summary.txt
