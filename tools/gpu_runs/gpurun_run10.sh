mkdir -p gpurun_out
timeout 1200 python tools/tune_pipe.py --rounds 5 --iters 5 --variants 4:8:1:256:1:2,2:8:1:256:1:1,1:4:1:256:1:4,1:2:1:256:1:4,1:8:1:256:1:4,4:8:1:256:1:4,1:4:1:128:1:4,1:4:1:64:1:4,1:4:1:256:0:4,2:8:1:256:1:3 > gpurun_out/tune5.log 2>&1
grep -v stream_read gpurun_out/tune5.log | tail -14
