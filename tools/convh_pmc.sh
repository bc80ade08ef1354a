# Counters of the bf16 fused convolution (conv_os5h_kernel) on its two reference shapes, separate --pmc passes
# (MI355X_MICROARCH.md: SQ 8 slots, TCC 4 -- FETCH_SIZE and WRITE_SIZE cannot share a pass):
#   bash tools/convh_pmc.sh <out-prefix> [kernel-name filter, default conv_os5h]   ->  gpurun_out/<out-prefix>.txt (+ .err with the
#   timing lines). PCS_PMC_SHAPES="level cin cout;..." picks the shapes, PCS_CONVH_WS=1 the weight-stationary kernel (filter conv_os6h),
#   PCS_PMC_PASSES="1 3 5 6" a subset of the passes.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
: > $R/gpurun_out/$1.txt; : > $R/gpurun_out/$1.err
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM" \
           "GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_WAVE32_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1)); rm -rf /tmp/hp$i
  if [ -n "$PCS_PMC_PASSES" ] && ! echo " $PCS_PMC_PASSES " | grep -q " $i "; then continue; fi
  timeout 300 rocprofv3 --pmc $set --output-format csv -d /tmp/hp$i -- python $R/tools/convh_pmc_bench.py 3 > /tmp/hp$i.log 2>&1
  f=$(find /tmp/hp$i -name "*counter_collection.csv" | head -1)
  echo "== pass $i: $set" >> $R/gpurun_out/$1.txt
  [ -n "$f" ] && python $R/tools/pmc_summary.py "$f" ${2:-conv_os5h} >> $R/gpurun_out/$1.txt
  grep -h "convh level" /tmp/hp$i.log >> $R/gpurun_out/$1.err; tail -1 /tmp/hp$i.log >> $R/gpurun_out/$1.err
done
