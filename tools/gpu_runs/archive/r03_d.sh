#!/bin/bash
# round 3, pass D: the whole GPU suite + smoke + the default bench line
export TMPDIR=/tmp
O=gpurun_out/${1:-r03d}; mkdir -p $O
( time timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1 ) 2>> $O/summary.txt; echo "pytest gpu rc=$?" >> $O/summary.txt
tail -5 $O/pytest_gpu.txt >> $O/summary.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?" >> $O/summary.txt
( time timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err ) 2>> $O/summary.txt; echo "bench rc=$?" >> $O/summary.txt
cat $O/summary.txt
