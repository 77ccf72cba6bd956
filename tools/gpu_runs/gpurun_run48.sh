mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
i=0
for pm in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_WAIT_ANY"; do
  i=$((i+1))
  timeout 240 rocprofv3 --kernel-trace --pmc $pm -d gpurun_out/pmc8_$i -o h -f csv -- python bench.py --density-q16 200 --no-cpu --steps 3 --warmup 1 > gpurun_out/pmc8_$i.log 2>&1
  f=gpurun_out/pmc8_$i/h_counter_collection.csv
  if [ -f $f ]; then (head -1 $f; grep "k_pipe_counts" $f | head -60) > gpurun_out/pmc8_$i.csv; fi
  rm -rf gpurun_out/pmc8_$i
done
ls -la gpurun_out | grep pmc8
