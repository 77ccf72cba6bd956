mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu19.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu19.log
tail -12 gpurun_out/pytest_gpu19.log
timeout 600 python tools/bench_or_sharded.py > gpurun_out/or_sharded_1.log 2>&1; tail -1 gpurun_out/or_sharded_1.log
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 tools/bench_or_sharded.py > gpurun_out/or_sharded_dist1.log 2>&1; tail -1 gpurun_out/or_sharded_dist1.log
