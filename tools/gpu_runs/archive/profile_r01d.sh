set -x
mkdir -p gpurun_out/r01d
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
run() { # tag, command...
  tag=$1; shift
  rm -rf gpurun_out/prof_$tag
  timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o $tag -f csv -- "$@" > gpurun_out/r01d/$tag.out 2> gpurun_out/r01d/$tag.err
  f=$(find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f gpurun_out/r01d/kernel_stats_$tag.csv
  rm -rf gpurun_out/prof_$tag
}
run headline python bench.py --no-cpu
run mixed1pct python bench.py --no-cpu --density-q16 328 --steps 10 --warmup 2
run configs24 python tools/bench_configs.py --which 2,4
run config5 python tools/bench_configs.py --which 5
run shift python tools/bench_shift.py
ls -la gpurun_out/r01d
head -3 gpurun_out/r01d/kernel_stats_headline.csv
