cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_fuse.py tests/test_range_sample.py -m gpu -q > gpurun_out/e_tests1.log 2>&1; echo "tests1 rc=$?"; tail -4 gpurun_out/e_tests1.log
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q -k "device_input or weight_prep or quantize" > gpurun_out/e_tests3.log 2>&1; echo "tests3 rc=$?"; tail -4 gpurun_out/e_tests3.log
timeout 900 python -m pytest tests/test_fullsize_parity.py -m gpu -q -k "fused or config4" > gpurun_out/e_tests2.log 2>&1; echo "tests2 rc=$?"; tail -4 gpurun_out/e_tests2.log
timeout 900 python tools/modelbench.py cylinder:reference,cylinder:fuse,spvcnn18:fuse,rpvnet34:fuse,minkunet18:reference,minkunet18:fuse,minkunet18:workload > gpurun_out/e_modelbench.json 2> gpurun_out/e_modelbench.err; cat gpurun_out/e_modelbench.json
timeout 600 python bench.py --no-cpu-baseline --models none --no-split-line > gpurun_out/e_bench.log 2> gpurun_out/e_bench.err; tail -1 gpurun_out/e_bench.log
