mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_s
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_s -o s -f csv -- python bench.py --no-cpu --steps 3 --warmup 1 > gpurun_out/prof_s.out 2>&1
f=$(find gpurun_out/prof_s -name "*kernel_stats.csv" | head -1); grep -E "k_pipe_sort|k_pipe_counts" $f | sed "s/(.*)\"/\"/"; rm -rf gpurun_out/prof_s
