#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r05_l; rm -rf $O; mkdir -p $O
timeout 900 python tools/bench_select.py > $O/select_table.jsonl 2> $O/err.txt
cat $O/select_table.jsonl
