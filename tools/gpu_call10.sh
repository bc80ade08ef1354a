mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/gputest.log 2>&1; tail -4 gpurun_out/gputest.log
bash tools/profile_bench.sh r3f > gpurun_out/r3f_profile_summary.txt 2>&1
timeout 400 python tools/hbm_kernel_table.py > gpurun_out/r3f_hbm_ops.md 2>&1
grep -o '"value": [0-9.]*' gpurun_out/r3f_bench.log
