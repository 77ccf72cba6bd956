#!/bin/bash
# round 4, run y: A/B on ONE box -- the committed gather form of k_agg_or_rows vs the slotted-slab variant (profiles/r04_cold/
# slotted_slab_variant.patch built as libbmx_slotted_variant.so), interleaved
export TMPDIR=/tmp
O=gpurun_out/${1:-r04y}; rm -rf $O; mkdir -p $O
for rep in 1 2 3; do
  for lib in libbmx.so libbmx_slotted_variant.so; do
    echo "lib $lib" | tee -a $O/ab.jsonl
    BMX_LIB=bitmagic_amd/lib/$lib timeout 300 python tools/tail_probe.py 4360 2>> $O/err.txt | tee -a $O/ab.jsonl
  done
done
echo "lib libbmx_slotted_variant.so nt" | tee -a $O/ab.jsonl
BMX_OR_NT=1 BMX_LIB=bitmagic_amd/lib/libbmx_slotted_variant.so timeout 300 python tools/tail_probe.py 4360 2>> $O/err.txt | tee -a $O/ab.jsonl
