#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r05_i; rm -rf $O; mkdir -p $O
for d in 0 2; do BMX_DIAG_C2=$d python tools/prof_prepare_or.py >> $O/diag.txt 2>&1; done
grep build_ms $O/diag.txt
