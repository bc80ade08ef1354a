mkdir -p gpurun_out
timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | grep -v Warning | grep -E "^E  |Error|passed|failed|^FAILED|^tests.*(FAILED|ERROR)" | head -80 > gpurun_out/r6_gputest_ws.txt
cat gpurun_out/r6_gputest_ws.txt
