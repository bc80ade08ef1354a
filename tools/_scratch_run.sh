mkdir -p gpurun_out
timeout 900 python bench.py --amp bf16 --device-input --no-cpu-baseline --models none > gpurun_out/r6_bench_bf16_di.log 2> gpurun_out/r6_bench_bf16_di.err; tail -1 gpurun_out/r6_bench_bf16_di.log | cut -c1-1500
PCS_CONVH_WS=0 timeout 900 python bench.py --amp bf16 --no-cpu-baseline --models none --no-device-input-line > gpurun_out/r6_bench_bf16_ws0.log 2>&1; tail -1 gpurun_out/r6_bench_bf16_ws0.log | cut -c1-600
PCS_PROFILE_STEADY=1 bash tools/profile_bench.sh round6_amp_bf16 --amp bf16 | tail -30
