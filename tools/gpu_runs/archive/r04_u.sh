#!/bin/bash
# round 4, run u: tail of the row kernel's grid (4360 tiles = 17.03 rounds of 256 workgroups)
export TMPDIR=/tmp
O=gpurun_out/${1:-r04u}; rm -rf $O; mkdir -p $O
timeout 600 python tools/tail_probe.py 2>&1 | tee $O/tail_probe.jsonl
