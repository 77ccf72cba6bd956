mkdir -p gpurun_out
timeout 600 python tools/ramp_probe.py > gpurun_out/ramp.log 2>&1
cat gpurun_out/ramp.log | tail -50
