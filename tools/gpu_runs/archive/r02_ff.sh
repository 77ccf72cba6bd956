#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/${1:-r02k}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "find_first or range_hint or small_collection or sparse_state or golden or scanner" > $O/pytest_ff.log 2>&1; echo "pytest rc=$?" >> $O/pytest_ff.log
tail -15 $O/pytest_ff.log
timeout 600 python tools/bench_small.py > $O/small.json 2>$O/small.err; cat $O/small.json; tail -3 $O/small.err
