#!/bin/bash
# GPU pass: tests, headline bench (+ all-cores reference baseline), the RCCL path through torch.distributed.run
set -x
export TMPDIR=/tmp
O=gpurun_out/${1:-r02}; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
cat $O/bench.json
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu > $O/bench_torchrun1.json 2> $O/bench_torchrun1.err; echo "torchrun rc=$?"
cat $O/bench_torchrun1.json
