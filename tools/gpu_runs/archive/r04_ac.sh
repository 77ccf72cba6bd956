#!/bin/bash
# round 4, run ac: same-box A/B of two library builds on the materialised pairwise operations (tools/op2_ab.py; BMX_LIB selects the build)
export TMPDIR=/tmp
O=gpurun_out/${1:-r04ac}; rm -rf $O; mkdir -p $O
for rep in 1 2; do for lib in libbmx_prev.so libbmx.so; do
  BMX_LIB=bitmagic_amd/lib/$lib python tools/op2_ab.py 655 2>/dev/null | tee -a $O/ab.jsonl
done; done
for lib in libbmx_prev.so libbmx.so; do BMX_LIB=bitmagic_amd/lib/$lib python tools/op2_ab.py 6554 2>/dev/null | tee -a $O/ab.jsonl; done
