#!/bin/bash
# Round 6: the whole -m gpu suite and both randomized soaks on red-zone contexts (BMX_DEBUG_REDZONE=1: 4 KiB canaries around every
# device allocation, verified at free / synchronize / destroy; a damaged zone makes bmx_ctx_synchronize fail and is reported on
# stderr as "[bmx redzone] ...").  Output -> gpurun_out/r06_redzone/ (copied to profiles/r06_redzone/).
out=gpurun_out/r06_redzone; mkdir -p $out
export BMX_DEBUG_REDZONE=1
python -m pytest tests -q -m gpu > $out/pytest.txt 2>&1
python tools/soak_r05.py 120 > $out/soak_r05.txt 2>&1
python tools/soak_r04.py 40 > $out/soak_r04.txt 2>&1
{ echo "red-zone reports in the three logs (the two lines of the checker's self-test inside test_red_zone_allocator... are expected in none of them: that test runs its own subprocess):";
  grep -c "bmx redzone" $out/pytest.txt $out/soak_r05.txt $out/soak_r04.txt; tail -3 $out/pytest.txt; tail -2 $out/soak_r05.txt; tail -2 $out/soak_r04.txt; } > $out/summary.txt
cat $out/summary.txt
