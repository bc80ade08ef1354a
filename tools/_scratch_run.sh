mkdir -p gpurun_out
timeout 900 python tools/wgrad_interleave_ab.py 2>&1 | grep -v amdgpu.ids > gpurun_out/wgrad_interleave_ab3.txt; cat gpurun_out/wgrad_interleave_ab3.txt
timeout 1500 python -m pytest tests/test_dense_parity.py tests/test_hip_parity.py -m gpu -x -q -k "wgrad or backward" 2>&1 | grep -E "^E  |passed|failed|^FAILED" | head
timeout 1500 python bench.py > gpurun_out/round6_default_bench.log 2> gpurun_out/round6_default_bench.err; tail -1 gpurun_out/round6_default_bench.log | cut -c1-2300
