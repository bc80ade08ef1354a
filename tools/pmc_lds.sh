# LDS counters (bank / address conflicts, busy cycles) of the fused conv on two shapes: bash tools/pmc_lds.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for shape in "0 96 96" "3 256 256"; do
rm -rf /tmp/pl; rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_UNALIGNED_STALL --output-format csv -d /tmp/pl -- python $R/tools/conv_microbench.py $shape 3 > /tmp/pl.log 2>&1
f=$(find /tmp/pl -name "*counter_collection.csv" | head -1)
echo "== $shape"; [ -n "$f" ] && python $R/tools/pmc_summary.py "$f" conv_os5 || tail -3 /tmp/pl.log
done
