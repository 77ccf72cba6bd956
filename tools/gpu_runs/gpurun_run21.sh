mkdir -p gpurun_out
timeout 600 python bench.py --independent --no-cpu > gpurun_out/bench_indep.json 2> gpurun_out/bench_indep.err
python -c "
import json;d=json.load(open('gpurun_out/bench_indep.json'));print('independent:', d['ms_per_step'], d['value'], d['config']['result_count'])"
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof5
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof5 -o c5 -f csv -- python tools/bench_configs.py --which 5 --or-vecs 4096 > gpurun_out/config5_prof.log 2>&1
rm -f gpurun_out/prof5/c5_kernel_trace.csv
grep -E "k_agg_or|k_or_sort|k_scan_layout|k_emit" gpurun_out/prof5/c5_kernel_stats.csv | cut -c1-200
