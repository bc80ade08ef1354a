cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/prof && PCS_BENCH_PREHEAT=0 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof19.log 2>&1
f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1); cp "$f" $R/gpurun_out/r19_kernel_stats.csv
cd $R && timeout 200 python bench.py > gpurun_out/bench19.log 2>&1; tail -1 gpurun_out/bench19.log | cut -c1-300
