mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 -L > gpurun_out/counters_list.txt 2>&1
grep -ciE "utcl|tlb" gpurun_out/counters_list.txt
for pm in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum"; do
  tag=$(echo $pm | cut -d' ' -f1)
  timeout 600 rocprofv3 --kernel-trace --pmc $pm -d gpurun_out/pmc5_$tag -o c5 -f csv -- python tools/bench_configs.py --which 5 --or-vecs 4096 > gpurun_out/pmc5_$tag.log 2>&1
  f=gpurun_out/pmc5_$tag/c5_counter_collection.csv
  if [ -f $f ]; then (head -1 $f; grep k_agg_or $f) > gpurun_out/pmc5_$tag.csv; fi
  rm -rf gpurun_out/pmc5_$tag
done
ls gpurun_out | grep pmc5
