#!/bin/bash
# everything DESIGN.md section 7 quotes, in one pass: GPU tests, headline (+ CPU legs), the RCCL path, configs 1/3/4,
# GAP-heavy variants of the headline, small collections
export TMPDIR=/tmp
O=gpurun_out/${1:-r02h}; mkdir -p $O
python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed" | tee $O/pytest.txt
python bench.py > $O/bench.json 2> $O/bench.err; cat $O/bench.json
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu > $O/bench_torchrun1.json 2> $O/bench_torchrun1.err
for c in 1 3 4; do python bench.py --config $c > $O/bench_config$c.json 2> /dev/null; cat $O/bench_config$c.json; done
python bench.py --config 1 --density-q16 655 --no-cpu > $O/bench_config1_1pct.json 2>/dev/null; cat $O/bench_config1_1pct.json
python bench.py --config 3 --density-q16 655 --no-cpu > $O/bench_config3_1pct.json 2>/dev/null; cat $O/bench_config3_1pct.json
for dq in 328 197 66; do python bench.py --density-q16 $dq --no-cpu --no-shard-probe > $O/bench_dq$dq.json 2>/dev/null; cat $O/bench_dq$dq.json; done
python bench.py --independent --no-cpu --no-shard-probe > $O/bench_indep.json 2>/dev/null; cat $O/bench_indep.json
python tools/bench_small.py > $O/bench_small.log 2>/dev/null; cat $O/bench_small.log
python tools/bench_scanner.py > $O/bench_scanner.log 2>/dev/null; tail -5 $O/bench_scanner.log
python tools/bench_shift.py > $O/bench_shift.log 2>/dev/null; tail -8 $O/bench_shift.log
