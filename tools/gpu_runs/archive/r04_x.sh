#!/bin/bash
# round 4, run x: k_agg_or_rows with pieces of its epilogue compiled out (tuning build: BMX_DIAG_ROWS 1024 = no folds, 2048 = nothing
# classified / stored, 512 = loads only) -- where do the 0.8 ms between the kernel and pieces_probe + row arithmetic go?
export TMPDIR=/tmp
O=gpurun_out/${1:-r04x}; rm -rf $O; mkdir -p $O
export BMX_LIB=bitmagic_amd/lib/libbmx_tune.so
for d in 0 1024 2048 3072 512 3584 0; do
  BMX_DIAG_ROWS=$d timeout 300 python tools/tail_probe.py 4360 2>> $O/err.txt | tee -a $O/rows_diag.jsonl
done
