#!/bin/bash
# full GPU suite + small-collection latency (direct one-launch aggregation, tools/bench_small.py)
set -x
export TMPDIR=/tmp
O=gpurun_out/${1:-r02i}; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -8 $O/pytest.log
timeout 300 python tools/bench_small.py > $O/small.json 2>$O/small.err; cat $O/small.json
BMX_DIRECT_COLS=0 timeout 300 python tools/bench_small.py > $O/small_nodirect.json 2>>$O/small.err; cat $O/small_nodirect.json
