#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r6_gputest2.txt 2>&1
tail -5 gpurun_out/r6_gputest2.txt
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --models none > gpurun_out/r6_bench_order.log 2>&1
tail -2 gpurun_out/r6_bench_order.log | cut -c1-1200
