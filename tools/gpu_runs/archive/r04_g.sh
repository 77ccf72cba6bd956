#!/bin/bash
# round 4, run g: kernel times of the first-call combine_or (configs[4], no packed collection)
export TMPDIR=/tmp
O=gpurun_out/${1:-r04g}; rm -rf $O; mkdir -p $O
export BMX_GAP_PACK=0
timeout 600 rocprofv3 --kernel-trace --stats -d $O/p -o x -f csv -- python bench.py --config 4 --no-cpu --steps 6 --warmup 2 > $O/bench.json 2> $O/err.txt
f=$(find $O/p -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_config4.csv && head -8 $O/kernel_stats_config4.csv | cut -c1-200
rm -rf $O/p
