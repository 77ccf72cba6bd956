#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r05_m; rm -rf $O; mkdir -p $O
( time timeout 2400 python -m pytest tests -q -m gpu -x ) > $O/pytest.txt 2>&1; echo "pytest gpu rc=$?" >> $O/summary.txt
tail -8 $O/pytest.txt >> $O/summary.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?" >> $O/summary.txt
