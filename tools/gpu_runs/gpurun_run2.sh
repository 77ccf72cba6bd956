set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r01 -o r01 -f csv -- python bench.py --steps 10 --warmup 2 --no-cpu > gpurun_out/prof_bench.json 2> gpurun_out/prof_bench.err
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_fetch -o r01 -f csv -- python bench.py --steps 3 --warmup 1 --no-cpu > gpurun_out/pmc_fetch.json 2> gpurun_out/pmc_fetch.err
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_write -o r01 -f csv -- python bench.py --steps 3 --warmup 1 --no-cpu > gpurun_out/pmc_write.json 2> gpurun_out/pmc_write.err
timeout 900 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d gpurun_out/pmc_tcc -o r01 -f csv -- python bench.py --steps 3 --warmup 1 --no-cpu > gpurun_out/pmc_tcc.json 2> gpurun_out/pmc_tcc.err
find gpurun_out -name "*.csv" | head -30; du -sh gpurun_out
# keep the output small: drop raw per-dispatch traces of the generator kernels except k_pipe rows
for d in prof_r01 pmc_fetch pmc_write pmc_tcc; do
  for f in $(find gpurun_out/$d -name "*kernel_trace.csv" -o -name "*counter_collection.csv"); do
    (head -1 $f; grep k_pipe_counts $f) > $f.pipe.csv; rm $f
  done
done
du -sh gpurun_out
