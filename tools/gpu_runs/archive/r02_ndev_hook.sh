#!/bin/bash
# the N > 1 code path of bench.py on a ONE-GPU box (test hook: every rank on device 0, gloo): sharded generation, per-rank
# gathers, the weak-scaling second figure and the JSON line; the timings of such a run mean nothing
export TMPDIR=/tmp
O=gpurun_out/${1:-r02p}; mkdir -p $O
for n in 2 4; do
  BMX_BENCH_TEST_ONE_DEVICE=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2951$n \
      bench.py --gpus $n --steps 5 --warmup 1 --nvec 64 > $O/bench_hook_n$n.json 2> $O/bench_hook_n$n.err; echo "n=$n rc=$?"
  python - $O/bench_hook_n$n.json <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: j[k] for k in ("n_gpus", "scaling", "value", "ms_per_step")}, j["config"]["result_count"], j["config"]["blocks_per_rank"], j["per_rank"], j.get("weak_scaling"))
PY
done
timeout 600 python bench.py --nvec 64 --no-cpu --no-shard-probe 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('single', j['config']['result_count'])"
BMX_BENCH_TEST_ONE_DEVICE=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 2 --config 4 --steps 3 --warmup 1 --no-cpu > $O/bench_hook_c4.json 2> $O/bench_hook_c4.err; echo "config4 n=2 rc=$?"; tail -c 600 $O/bench_hook_c4.json
