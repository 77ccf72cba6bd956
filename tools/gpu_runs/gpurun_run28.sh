mkdir -p gpurun_out
timeout 1200 python tools/tune_pipe.py --rounds 5 --iters 5 --variants 4:8:1:256:1:2:0,4:8:1:512:1:2:0,4:8:1:768:1:2:0,4:8:1:384:1:2:0,4:8:1:256:1:2:81920,4:8:1:256:1:2:55000,4:8:1:256:1:2:163000,4:8:1:512:1:2:81920,4:8:1:128:1:2:40960 > gpurun_out/tune8.log 2>&1
grep -E "^ver|rror" gpurun_out/tune8.log
