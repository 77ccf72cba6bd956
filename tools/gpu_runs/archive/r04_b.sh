#!/bin/bash
# round 4, run b: the load side of the tiled OR kernel rebuilt step by step
export TMPDIR=/tmp
O=gpurun_out/${1:-r04b}; mkdir -p $O
timeout 600 tools/bin/tiled_probe > $O/tiled_probe.jsonl 2> $O/tiled_probe.err
cat $O/tiled_probe.jsonl; tail -3 $O/tiled_probe.err
