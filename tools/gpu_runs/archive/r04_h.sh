#!/bin/bash
# round 4, run h: packed collections as first-class objects (member directory): parity
export TMPDIR=/tmp
O=gpurun_out/${1:-r04h}; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "row_kernel or many_gap_operands or packed or prepared_collection" > $O/pytest.log 2>&1
tail -25 $O/pytest.log
