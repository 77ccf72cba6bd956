#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r05_f; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_cpp_facade.py -q -x -k "op2_count or pairwise or golden_case or async or facade or edge" > $O/pytest_sel.txt 2>&1; echo "pytest rc=$?" >> $O/summary.txt
tail -6 $O/pytest_sel.txt >> $O/summary.txt
timeout 300 python bench.py --config 0 > $O/bench_config0.json 2>> $O/err.txt
python -c "
import json; r = json.load(open('$O/bench_config0.json')); print('config0', r['value'], r['ms_per_step'], r.get('cpu_baseline', {}).get('value'))" >> $O/summary.txt
