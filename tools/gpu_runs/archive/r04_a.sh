#!/bin/bash
# round 4, run a: what bounds the cold (descriptor-table) combine_or? access-pattern probe
export TMPDIR=/tmp
O=gpurun_out/${1:-r04a}; mkdir -p $O
timeout 600 tools/bin/pieces_probe > $O/pieces_probe.jsonl 2> $O/pieces_probe.err
cat $O/pieces_probe.jsonl; tail -3 $O/pieces_probe.err
