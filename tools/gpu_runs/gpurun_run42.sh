mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
i=0
for pm in "FETCH_SIZE WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_ANY SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $pm -d gpurun_out/pmc7_$i -o h -f csv -- python tools/bench_configs.py --which 5 --or-vecs 4096 > gpurun_out/pmc7_$i.log 2>&1
  f=gpurun_out/pmc7_$i/h_counter_collection.csv
  if [ -f $f ]; then (head -1 $f; grep k_agg_or $f | head -40) > gpurun_out/pmc7_$i.csv; fi
  rm -rf gpurun_out/pmc7_$i
done
ls -la gpurun_out | grep pmc7
