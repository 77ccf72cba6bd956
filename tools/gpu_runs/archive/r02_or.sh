#!/bin/bash
# configs[4]: memory-pattern probes (loads only, tuning build) vs the real kernels
export TMPDIR=/tmp
export BMX_LIB=$PWD/bitmagic_amd/lib/libbmx_tune.so
O=gpurun_out/${1:-r02e}; mkdir -p $O
for v in 0 1 2; do for w in 0 -9; do
  BMX_OR_TILE=$v BMX_OR_WINDOW=$w python bench.py --config 4 --no-cpu --steps 10 > $O/or_tile${v}_$w.json 2>/dev/null
  python -c "import json; j=json.load(open('$O/or_tile${v}_$w.json')); print('or_tile=$v probe=$w', j['ms_per_step'], j['roofline']['frac'], j['config']['result_count'])"
done; done
