cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_dense_parity.py -m gpu -q -k "half_conv_backward" > gpurun_out/f_tests1.log 2>&1; echo "tests1 rc=$?"; tail -3 gpurun_out/f_tests1.log
PCS_WGRAD3_THIN=0 timeout 300 python tools/wgrad_thin_ab.py 2>&1 | tail -6
timeout 300 python tools/wgrad_thin_ab.py 2>&1 | tail -6
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q -k "device_input" > gpurun_out/f_tests2.log 2>&1; echo "tests2 rc=$?"; tail -3 gpurun_out/f_tests2.log
timeout 900 python -m pytest tests/test_fullsize_parity.py -m gpu -q -k "fused and (config4 or config5)" > gpurun_out/f_tests3.log 2>&1; echo "tests3 rc=$?"; tail -3 gpurun_out/f_tests3.log
timeout 600 python bench.py --no-cpu-baseline --models none --no-split-line --amp bf16 --device-input > gpurun_out/f_bench.log 2> gpurun_out/f_bench.err; tail -1 gpurun_out/f_bench.log | cut -c1-200; tail -1 gpurun_out/f_bench.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d.get('device_input'), d['value'])"
