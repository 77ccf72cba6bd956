mkdir -p gpurun_out
export MASTER_ADDR=127.0.0.1
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 10 --warmup 2 > gpurun_out/bench_torchrun1.json 2> gpurun_out/bench_torchrun1.err
echo "rc=$?"; tail -3 gpurun_out/bench_torchrun1.err; cut -c1-300 gpurun_out/bench_torchrun1.json
