mkdir -p gpurun_out
timeout 1200 python tools/tune_pipe.py --rounds 5 --iters 5 --variants 4:8:1:256:1:2,4:8:1:512:1:2,2:8:1:512:1:2,2:8:1:1024:1:2,2:8:1:256:1:2,4:8:1:128:1:2,4:8:1:64:1:2 > gpurun_out/tune7.log 2>&1
grep -E "^ver" gpurun_out/tune7.log
