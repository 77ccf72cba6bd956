mkdir -p gpurun_out
timeout 1200 python tools/tune_pipe.py --rounds 5 --iters 5 > gpurun_out/tune3.log 2>&1
grep -v stream_read gpurun_out/tune3.log | tail -16
