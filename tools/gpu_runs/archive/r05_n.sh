#!/bin/bash
# pairwise results with GAP blocks: k_op2_loop converts its own GAP candidates (tail phase) -- tests, soak, bench of configs[1] at 1 %
export TMPDIR=/tmp
O=gpurun_out/r05_n; rm -rf $O; mkdir -p $O
( time timeout 1200 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "pairwise or async or op2 or golden or bm64 or adapter" ) > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/summary.txt
tail -5 $O/pytest.txt >> $O/summary.txt
timeout 600 python tools/soak_r04.py 30 > $O/soak_r04.txt 2>&1; echo "soak_r04 rc=$?" >> $O/summary.txt; tail -3 $O/soak_r04.txt >> $O/summary.txt
timeout 300 python bench.py --config 1 --density-q16 655 > $O/bench_config1_1pct.json 2> $O/bench_config1_1pct.err; echo "bench rc=$?" >> $O/summary.txt
python - <<'PY' >> $O/summary.txt
import json
d=json.loads(open('gpurun_out/r05_n/bench_config1_1pct.json').read().strip().splitlines()[-1])
print(json.dumps(d['config']['per_op']))
PY
cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/ks -o s -f csv -- python $GRAFT_REPO_ROOT/bench.py --config 1 --density-q16 655 > /dev/null 2>&1; cd $GRAFT_REPO_ROOT
cp /tmp/ks/*kernel_stats.csv $O/kernel_stats_config1_1pct.csv 2>/dev/null || find /tmp/ks -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_config1_1pct.csv \;
grep -i "op2_loop\|emit_gaps\|rebase\|copyBuffer" $O/kernel_stats_config1_1pct.csv | cut -c1-60,200- >> $O/summary.txt
cat $O/summary.txt
