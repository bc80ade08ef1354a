# rocprofv3 --kernel-trace --stats of ONE tools/modelbench.py record (the reference's segmentor sources on the HIP backend),
# steady state = difference of a (2 + 6)-step and a (2 + 2)-step trace = 4 training steps:
#   bash tools/profile_model.sh <tag> <spec, e.g. minkunet34:fuse:bf16>  ->  gpurun_out/<tag>_steady4_kernel_stats.csv, <tag>_step_budget.md
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; SPEC=$2
for N in 2 6; do
  rm -rf /tmp/pm_${TAG}_$N
  PCS_BENCH_PREHEAT=0 PCS_MB_STEPS=$N PCS_MB_WARMUP=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pm_${TAG}_$N -- python $R/tools/modelbench.py $SPEC > $R/gpurun_out/${TAG}_prof$N.log 2>&1
done
f2=$(find /tmp/pm_${TAG}_2 -name "*kernel_stats.csv" | head -1); f6=$(find /tmp/pm_${TAG}_6 -name "*kernel_stats.csv" | head -1)
python $R/tools/stats_diff.py "$f2" "$f6" $R/gpurun_out/${TAG}_steady4_kernel_stats.csv
python $R/tools/step_budget.py $R/gpurun_out/${TAG}_steady4_kernel_stats.csv 4 "$TAG ($SPEC): steady-state step (difference of two traces, 4 steps)" > $R/gpurun_out/${TAG}_step_budget.md
cat $R/gpurun_out/${TAG}_step_budget.md
tail -1 $R/gpurun_out/${TAG}_prof6.log | cut -c1-300
