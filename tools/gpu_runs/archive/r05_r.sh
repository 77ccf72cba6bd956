#!/bin/bash
# k_op2_loop as one 16-wave workgroup per CU when GAP results can come out: tests, soak, configs[1] at 1 % with both shapes
export TMPDIR=/tmp
O=gpurun_out/r05_r; rm -rf $O; mkdir -p $O
( time timeout 1200 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "pairwise or async or op2 or golden or bm64 or adapter" ) > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/summary.txt
tail -5 $O/pytest.txt >> $O/summary.txt
timeout 600 python tools/soak_r04.py 30 > $O/soak_r04.txt 2>&1; echo "soak_r04 rc=$?" >> $O/summary.txt; tail -2 $O/soak_r04.txt >> $O/summary.txt
for w in 16 4; do
BMX_OP2_WAVES=$w timeout 300 python bench.py --config 1 --density-q16 655 > $O/bench_config1_1pct_waves$w.json 2> $O/err$w.txt; echo "bench waves $w rc=$?" >> $O/summary.txt
python - $w <<'PY' >> $O/summary.txt
import json, sys
d=json.loads(open('gpurun_out/r05_r/bench_config1_1pct_waves%s.json' % sys.argv[1]).read().strip().splitlines()[-1])
print({k: (v['materialised_host_call_ms'], v['materialised_async_ms_per_op']) for k, v in d['config']['per_op'].items()})
PY
done
timeout 120 python bench.py --config 0 > $O/bench_config0.json 2>> $O/err.txt
python - <<'PY' >> $O/summary.txt
import json
d=json.loads(open('gpurun_out/r05_r/bench_config0.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['config'].get('one_call_op2_count_ms'), d['config'].get('count_and_only_ms'), d['config']['counts_equal'])
PY
cat $O/summary.txt
# (the BMX_OP2_WAVES knob this script sets was removed after this run: the 16-wave shape lost, profiles/r05_pair/README.md)
