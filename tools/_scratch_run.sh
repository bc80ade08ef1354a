#!/bin/bash
R=/root/repo; mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_g && PCS_BENCH_PREHEAT=0 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_g -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-device-input-line --no-split-line --models none --amp bf16 > $R/gpurun_out/gaps_prof.log 2>&1
t=$(find /tmp/prof_g -name "*kernel_trace.csv" | head -1)
head -1 $t > $R/gpurun_out/gaps_header.txt
python $R/tools/step_gaps.py $t 3 5 > $R/gpurun_out/step_gaps_bf16.txt 2>&1
cat $R/gpurun_out/step_gaps_bf16.txt
cd $R; python tools/wgrad_microbench.py 2>/dev/null | tail -12
