#!/bin/bash
# round 4, run w: what the row arithmetic / the LDS atomics of k_agg_or_rows cost a pure 4096 x 1 KiB piece stream (pieces_probe work modes)
export TMPDIR=/tmp
O=gpurun_out/${1:-r04w}; rm -rf $O; mkdir -p $O
for w in 0 1 2 0; do
  echo "work $w" >> $O/pieces_work.jsonl
  timeout 300 tools/bin/pieces_probe 4096 3418016 1 1 $w 2>&1 | grep '"sep"' >> $O/pieces_work.jsonl
done
cat $O/pieces_work.jsonl
