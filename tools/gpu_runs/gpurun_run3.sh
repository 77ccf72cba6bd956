mkdir -p gpurun_out
timeout 1200 python tools/tune_pipe.py --rounds 5 --iters 5 > gpurun_out/tune1.log 2>&1
tail -40 gpurun_out/tune1.log
