mkdir -p gpurun_out
timeout 1200 python tools/tune_pipe.py --rounds 5 --iters 5 --variants 4:8:1:256:1:2,4:8:1:256:1:5,2:8:1:256:1:5,1:8:1:256:1:5,2:8:1:256:1:2 > gpurun_out/tune6.log 2>&1
tail -8 gpurun_out/tune6.log
