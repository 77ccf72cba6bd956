#!/bin/bash
# round 4, run ag: the whole GPU suite (after the C++ facade for the asynchronous operations)
export TMPDIR=/tmp
O=gpurun_out/${1:-r04ag}; rm -rf $O; mkdir -p $O
timeout 1800 python -m pytest tests -q -m gpu > $O/pytest.txt 2>&1; echo "rc $?" >> $O/pytest.txt; tail -5 $O/pytest.txt
