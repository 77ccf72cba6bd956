#!/bin/bash
# round 4, run ae: the whole GPU suite
export TMPDIR=/tmp
O=gpurun_out/${1:-r04ae}; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu -x > $O/pytest.txt 2>&1; echo "rc $?" >> $O/pytest.txt; tail -4 $O/pytest.txt
