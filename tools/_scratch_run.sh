mkdir -p gpurun_out
timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | grep -v Warning | grep -E "^E  |Error|passed|failed|^FAILED" | head -40 > gpurun_out/r6_gputest2.txt
cat gpurun_out/r6_gputest2.txt
for lk in 1 0; do
PCS_BN_BWD_LINK=$lk timeout 900 python bench.py --amp bf16 --no-cpu-baseline --models none --no-device-input-line > gpurun_out/r6_bench_bf16_link$lk.log 2>&1; tail -1 gpurun_out/r6_bench_bf16_link$lk.log | cut -c1-200
PCS_BN_BWD_LINK=$lk timeout 900 python bench.py --no-amp-line --no-split-line --no-cpu-baseline --models none --no-device-input-line > gpurun_out/r6_bench_f32_link$lk.log 2>&1; tail -1 gpurun_out/r6_bench_f32_link$lk.log | cut -c1-200
done
