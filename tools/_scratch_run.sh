#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r6_gputest3.txt 2>&1
tail -3 gpurun_out/r6_gputest3.txt
