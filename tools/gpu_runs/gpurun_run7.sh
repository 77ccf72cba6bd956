mkdir -p gpurun_out
timeout 1200 python tools/tune_pipe.py --rounds 5 --iters 5 --variants 2:8:1:256:1,2:8:0:256:1,4:8:1:256:1,1:8:1:256:1,2:8:1:128:1,2:8:1:64:1,2:8:1:256:0 > gpurun_out/tune2.log 2>&1
cat gpurun_out/tune2.log | tail -25
