#!/bin/bash
# quick GPU pass: the whole GPU suite only
export TMPDIR=/tmp
O=gpurun_out/${1:-r02q}; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -25 $O/pytest.log
