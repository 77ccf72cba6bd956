set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof3 gpurun_out/pmc3_*
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof3 -o r01c -f csv -- python bench.py --no-cpu > gpurun_out/prof3_bench.json 2> gpurun_out/prof3_bench.err
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc3_fetch -o r01c -f csv -- python bench.py --steps 3 --warmup 1 --no-cpu > /dev/null 2> gpurun_out/pmc3_fetch.err
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc3_write -o r01c -f csv -- python bench.py --steps 3 --warmup 1 --no-cpu > /dev/null 2> gpurun_out/pmc3_write.err
timeout 900 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d gpurun_out/pmc3_tcc -o r01c -f csv -- python bench.py --steps 3 --warmup 1 --no-cpu > /dev/null 2> gpurun_out/pmc3_tcc.err
timeout 900 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d gpurun_out/pmc3_sq -o r01c -f csv -- python bench.py --steps 3 --warmup 1 --no-cpu > /dev/null 2> gpurun_out/pmc3_sq.err
for d in prof3 pmc3_fetch pmc3_write pmc3_tcc pmc3_sq; do
  for f in $(find gpurun_out/$d -name "*kernel_trace.csv" -o -name "*counter_collection.csv"); do
    (head -1 $f; grep k_pipe_counts $f) > $f.pipe.csv; rm $f
  done
done
cat gpurun_out/prof3/r01c_kernel_stats.csv | head -4
