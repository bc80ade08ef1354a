# rocprofv3 --kernel-trace --stats of bench.py (3 steps in the trace) + the plain bench line of the same configuration:
#   bash tools/profile_bench.sh <tag> [bench args, e.g. --amp bf16]  ->  gpurun_out/<tag>_kernel_stats.csv, <tag>_bench.log
#   PCS_PROFILE_STEADY=1: also <tag>_steady4_kernel_stats.csv / <tag>_step_budget.md (set-up launches cancelled out)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; shift
rm -rf /tmp/prof_$TAG && PCS_BENCH_PREHEAT=0 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-device-input-line --models none "$@" > $R/gpurun_out/${TAG}_prof.log 2>&1
f=$(find /tmp/prof_$TAG -name "*kernel_stats.csv" | head -1); cp "$f" $R/gpurun_out/${TAG}_kernel_stats.csv
t=$(find /tmp/prof_$TAG -name "*kernel_trace.csv" | head -1); python $R/tools/kernel_shapes.py "$t" bn_ devoxelize lovasz > $R/gpurun_out/${TAG}_small_kernels.txt 2>&1
if [ "$PCS_PROFILE_STEADY" = "1" ]; then   # a second trace with 4 more timed steps: the difference is 4 steady-state steps
  rm -rf /tmp/prof2_$TAG && PCS_BENCH_PREHEAT=0 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof2_$TAG -- python $R/bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-device-input-line --models none "$@" > $R/gpurun_out/${TAG}_prof2.log 2>&1
  f2=$(find /tmp/prof2_$TAG -name "*kernel_stats.csv" | head -1)
  python $R/tools/stats_diff.py "$f" "$f2" $R/gpurun_out/${TAG}_steady4_kernel_stats.csv
  python $R/tools/step_budget.py $R/gpurun_out/${TAG}_steady4_kernel_stats.csv 4 "$TAG: steady-state step (difference of a 7-step and a 3-step trace)" > $R/gpurun_out/${TAG}_step_budget.md
fi
cd $R && timeout 400 python bench.py --no-device-input-line --models none "$@" > gpurun_out/${TAG}_bench.log 2> gpurun_out/${TAG}_bench.err; tail -1 gpurun_out/${TAG}_bench.log | cut -c1-400
python - <<PY
import csv
rows = list(csv.DictReader(open("$R/gpurun_out/${TAG}_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:24]:
    print("%-84s %5s %8.2f ms/step %5.1f%%" % (r["Name"][:84], r["Calls"], float(r["TotalDurationNs"]) / 3e6, 100 * float(r["TotalDurationNs"]) / tot))
print("kernels per step: %.1f ms" % (tot / 3e6))
PY
