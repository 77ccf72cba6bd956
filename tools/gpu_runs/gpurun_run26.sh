mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
i=0
for pm in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum" "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum TCC_BUSY_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" "TA_TA_BUSY_sum GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $pm -d gpurun_out/pmc6_$i -o h -f csv -- python bench.py --steps 3 --warmup 1 --no-cpu > gpurun_out/pmc6_$i.log 2>&1
  f=gpurun_out/pmc6_$i/h_counter_collection.csv
  if [ -f $f ]; then (head -1 $f; grep k_pipe_counts $f) > gpurun_out/pmc6_$i.csv; fi
  rm -rf gpurun_out/pmc6_$i
done
ls gpurun_out | grep pmc6
