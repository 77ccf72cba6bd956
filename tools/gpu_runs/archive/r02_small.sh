#!/bin/bash
# full GPU suite + small-collection latency (direct one-launch aggregation, tools/bench_small.py) + pairwise configs
set -x
export TMPDIR=/tmp
O=gpurun_out/${1:-r02i}; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -8 $O/pytest.log
timeout 300 python tools/bench_small.py > $O/small.json 2>$O/small.err; cat $O/small.json
timeout 300 python bench.py --config 1 --no-cpu > $O/config1.json 2>>$O/small.err; cat $O/config1.json
timeout 300 python bench.py --config 1 --density-q16 655 --no-cpu > $O/config1_1pct.json 2>>$O/small.err; cat $O/config1_1pct.json
