mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/gputest.log 2>&1; tail -6 gpurun_out/gputest.log
bash tools/profile_bench.sh r3 > gpurun_out/r3_profile_summary.txt 2>&1
bash tools/conv_traffic.sh > gpurun_out/r3_conv_traffic.log 2>&1
PCS_BWD_OVERLAP=1 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r3_bench_overlap.log 2> gpurun_out/r3_bench_overlap.err
timeout 300 python tools/conv_layer_table.py 12 1.0 off > gpurun_out/r3_layer_table.txt 2>&1
timeout 900 python bench.py --steps 10 --warmup 3 > gpurun_out/r3_bench_full.log 2> gpurun_out/r3_bench_full.err
grep -o '"value": [0-9.]*' gpurun_out/r3_bench.log gpurun_out/r3_bench_overlap.log gpurun_out/r3_bench_full.log
