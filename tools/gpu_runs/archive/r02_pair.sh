#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/${1:-r02l}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pairwise or golden" > $O/pytest_pair.log 2>&1; echo "pytest rc=$?" >> $O/pytest_pair.log
tail -12 $O/pytest_pair.log
for nb in 1000000000 4000000000; do
for cfg in "0 1 1" "-1 1 1" "-1 1 0" "-1 2 1"; do set -- $cfg; echo -n "nbits $nb pair_stream $1 wgs $2 nt $3: "; BMX_PAIR_STREAM=$1 BMX_PAIR_WGS=$2 BMX_PIPE_NT=$3 timeout 300 python bench.py --config 1 --no-cpu --nbits $nb 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print({k: (v['kernel_ms'], v['materialised_host_call_ms']) for k, v in j['config']['per_op'].items()}, j['roofline']['frac'])"; done; done
