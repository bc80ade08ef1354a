mkdir -p gpurun_out; : > gpurun_out/r3c8_x3_ablate.txt
for v in product xab1 xab2 xab3 xab4; do
  if [ $v = product ]; then unset PCS_LIB_PATH; else export PCS_LIB_PATH=$PWD/openpcseg_amd/lib/dbg/$v.so; fi
  echo "== $v" >> gpurun_out/r3c8_x3_ablate.txt
  timeout 300 python tools/convx_pmc_bench.py 40 60 >> gpurun_out/r3c8_x3_ablate.txt 2>&1
done
unset PCS_LIB_PATH
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
: > $R/gpurun_out/r3c8_x3_pmc.txt
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM" \
           "GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1)); rm -rf /tmp/xp$i
  timeout 300 rocprofv3 --pmc $set --output-format csv -d /tmp/xp$i -- python $R/tools/convx_pmc_bench.py 3 1 > /tmp/xp$i.log 2>&1
  f=$(find /tmp/xp$i -name "*counter_collection.csv" | head -1)
  echo "== pass $i: $set" >> $R/gpurun_out/r3c8_x3_pmc.txt
  [ -n "$f" ] && python $R/tools/pmc_summary.py "$f" conv_os5x >> $R/gpurun_out/r3c8_x3_pmc.txt
done
grep -v amdgpu $R/gpurun_out/r3c8_x3_ablate.txt
