#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_reference_models.py -m gpu -x -q 2>&1 | tail -4
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --models none --no-split-line --no-device-input-line > gpurun_out/r6_bench_su.log 2>&1
tail -1 gpurun_out/r6_bench_su.log | cut -c1-300; grep -o '"amp_bf16":{"value":[0-9.]*,"ms_per_step":[0-9.]*' gpurun_out/r6_bench_su.log
