mkdir -p gpurun_out
timeout 1200 python tools/tune_pipe.py --rounds 3 --iters 5 --variants 4:8:1:256:1:2,2:8:1:256:1:1 > gpurun_out/tune4.log 2>&1
tail -16 gpurun_out/tune4.log
