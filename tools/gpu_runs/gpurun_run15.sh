mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof5
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof5 -o c5 -f csv -- python tools/bench_configs.py --which 5 --or-vecs 4096 > gpurun_out/config5_prof.log 2>&1
rm -f gpurun_out/prof5/c5_kernel_trace.csv
head -12 gpurun_out/prof5/c5_kernel_stats.csv
