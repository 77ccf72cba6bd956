#!/bin/bash
# round 4, run aj: prepared collections and the search count limit over device groups (bmx_gcollection_prepare, bmx_gpipeline_set_search_count_limit)
export TMPDIR=/tmp
O=gpurun_out/${1:-r04aj}; rm -rf $O; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_group.py -q -m gpu -x > $O/pytest_group.txt 2>&1; echo "rc $?" >> $O/pytest_group.txt; tail -12 $O/pytest_group.txt
