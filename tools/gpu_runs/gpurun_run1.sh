set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
rocm-smi --showproductname 2>&1 | head -8 > gpurun_out/smi.txt
nproc >> gpurun_out/smi.txt; grep -m1 "model name" /proc/cpuinfo >> gpurun_out/smi.txt; free -g | head -2 >> gpurun_out/smi.txt
timeout 900 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
timeout 600 python bench.py --steps 10 --warmup 2 > gpurun_out/bench_u2.json 2> gpurun_out/bench_u2.err
for u in 1 4; do BMX_PIPE_UNROLL=$u timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu > gpurun_out/bench_u$u.json 2> gpurun_out/bench_u$u.err; done
BMX_XCD_SWIZZLE=0 timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu > gpurun_out/bench_noswz.json 2> gpurun_out/bench_noswz.err
tail -3 gpurun_out/smoke.log; tail -15 gpurun_out/pytest_gpu.log; cat gpurun_out/bench_*.json
