#!/bin/bash
# round 3, pass A: group layer (workers, weighted shards), bench.py entry modes, config 3 with the gather probe + PMC
export TMPDIR=/tmp
O=gpurun_out/${1:-r03a}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_group.py -x -q -m gpu > $O/pytest_group.txt 2>&1; echo "pytest group rc=$?" >> $O/summary.txt
tail -3 $O/pytest_group.txt >> $O/summary.txt
( time timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err ) 2>> $O/summary.txt; echo "bench rc=$?" >> $O/summary.txt
BMX_BENCH_TEST_ONE_DEVICE=1 timeout 600 python bench.py --gpus 2 --nvec 32 --steps 5 --warmup 2 --no-cpu > $O/bench_group2_hook.json 2> $O/bench_group2.err; echo "group hook rc=$?" >> $O/summary.txt
timeout 300 python bench.py --gpus 2 --no-cpu --steps 2 > $O/bench_gpus2_refused.out 2> $O/bench_gpus2_refused.err; echo "plain --gpus 2 on one GPU rc=$? (must be non-zero)" >> $O/summary.txt
timeout 600 python bench.py --config 3 --steps 10 > $O/bench_config3.json 2> $O/bench_config3.err; echo "config3 rc=$?" >> $O/summary.txt
timeout 600 python bench.py --config 3 --density-q16 655 --steps 10 --no-allcores > $O/bench_config3_1pct.json 2>> $O/bench_config3.err
# PMC passes over k_rank / k_select / k_probe_lines
for set in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  rm -rf /tmp/pmc_c3
  timeout 600 rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc_c3 -o c3 -f csv -- python bench.py --config 3 --no-cpu --steps 3 --warmup 1 > /dev/null 2>> $O/pmc.err
  f=$(find /tmp/pmc_c3 -name "*counter_collection.csv" | head -1)
  python - "$f" >> $O/pmc_config3.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    for k in ("k_rank", "k_select", "k_probe_lines"):
        if r["Kernel_Name"].startswith(k): acc[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()): print(k[0], k[1], "per launch avg", sum(v) / len(v), "launches", len(v))
PY
done
cat $O/summary.txt; cat $O/pmc_config3.txt
