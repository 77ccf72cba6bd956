#!/bin/bash
# round 4, run j: bmx_collection_prepare kernel by kernel (rocprofv3 --stats over tools/prof_prepare.py; profiles/r04_coll/kernel_stats_prepare.csv)
export TMPDIR=/tmp
O=gpurun_out/${1:-r04j}; rm -rf $O; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats -d $O/p -o x -f csv -- python tools/prof_prepare.py > $O/out.txt 2> $O/err.txt
cat $O/out.txt
f=$(find $O/p -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_prepare.csv && grep -E "k_coll|copyBuffer|fillBuffer" $O/kernel_stats_prepare.csv | sed 's/(.*)"/"/' | cut -c1-150
rm -rf $O/p
