#!/bin/bash
# round 4, run an: workgroups per CU of the streaming materialising kernel (k_op2_stream) against its probe's best shape
export TMPDIR=/tmp
O=gpurun_out/${1:-r04an}; rm -rf $O; mkdir -p $O
for rep in 1 2; do for w in 4 8 6 2; do
  BMX_OP2_WGS=$w timeout 300 python tools/op2_ab.py 6554 2>/dev/null | sed "s/^/wgs=$w /" | tee -a $O/op2_wgs.txt
done; done
