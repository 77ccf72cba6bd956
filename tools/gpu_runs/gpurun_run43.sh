mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu17.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu17.log
tail -3 gpurun_out/pytest_gpu17.log
rm -rf gpurun_out/prof_c24
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_c24 -o c24 -f csv -- python tools/bench_configs.py --which 2,4 > gpurun_out/bench_c24.log 2>&1
f=$(find gpurun_out/prof_c24 -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/kernel_stats_c24.csv; rm -rf gpurun_out/prof_c24
grep -E "k_scan_layout|k_rs_scan|k_op2|k_rs_build" gpurun_out/kernel_stats_c24.csv | cut -c1-40,80-200
grep -E "bit_and|rs_index|build" gpurun_out/bench_c24.log | cut -c1-300 | head
