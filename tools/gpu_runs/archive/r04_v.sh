#!/bin/bash
# round 4, run v: randomized differential soak of this round's kernels (tools/soak_r04.py)
export TMPDIR=/tmp
O=gpurun_out/${1:-r04v}; rm -rf $O; mkdir -p $O
timeout 1100 python tools/soak_r04.py ${2:-240} > $O/soak_r04.log 2>&1; echo "rc $?" >> $O/soak_r04.log
tail -12 $O/soak_r04.log
