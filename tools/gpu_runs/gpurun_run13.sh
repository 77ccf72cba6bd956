set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof2 gpurun_out/pmc2_*
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof2 -o r01b -f csv -- python bench.py --no-cpu > gpurun_out/prof2_bench.json 2> gpurun_out/prof2_bench.err
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc2_fetch -o r01b -f csv -- python bench.py --steps 3 --warmup 1 --no-cpu > /dev/null 2> gpurun_out/pmc2_fetch.err
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc2_write -o r01b -f csv -- python bench.py --steps 3 --warmup 1 --no-cpu > /dev/null 2> gpurun_out/pmc2_write.err
timeout 900 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d gpurun_out/pmc2_tcc -o r01b -f csv -- python bench.py --steps 3 --warmup 1 --no-cpu > /dev/null 2> gpurun_out/pmc2_tcc.err
timeout 900 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d gpurun_out/pmc2_sq -o r01b -f csv -- python bench.py --steps 3 --warmup 1 --no-cpu > /dev/null 2> gpurun_out/pmc2_sq.err
for d in prof2 pmc2_fetch pmc2_write pmc2_tcc pmc2_sq; do
  for f in $(find gpurun_out/$d -name "*kernel_trace.csv" -o -name "*counter_collection.csv"); do
    (head -1 $f; grep k_pipe_counts $f) > $f.pipe.csv; rm $f
  done
done
cat gpurun_out/prof2/r01b_kernel_stats.csv | head -4
