mkdir -p gpurun_out
timeout 1200 python tools/tune_pipe.py --rounds 5 --iters 5 --variants 4:8:1:256:1:2:0,4:8:1:192:1:2:0,4:8:1:320:1:2:0,4:8:1:384:1:2:0,4:8:1:448:1:2:0,4:8:1:512:1:2:0,4:8:1:576:1:2:0,4:8:1:640:1:2:0,4:8:1:768:1:2:0,4:8:1:384:0:2:0 > gpurun_out/tune9.log 2>&1
grep -E "^ver|rror" gpurun_out/tune9.log
