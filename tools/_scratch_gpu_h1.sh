cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 bash tools/conv_traffic.sh > gpurun_out/h1_traffic.log 2>&1; tail -2 gpurun_out/h1_traffic.log | cut -c1-600
cp gpurun_out/conv_traffic.json profiles/round5_conv_traffic.json
PCS_PROFILE_STEADY=1 timeout 900 bash tools/profile_bench.sh round5 --no-amp-line --no-split-line > gpurun_out/h1_prof_f32.log 2>&1; tail -3 gpurun_out/h1_prof_f32.log
PCS_PROFILE_STEADY=1 timeout 900 bash tools/profile_bench.sh round5_amp_bf16 --amp bf16 > gpurun_out/h1_prof_bf16.log 2>&1; tail -3 gpurun_out/h1_prof_bf16.log
timeout 600 python -m pytest tests/test_fuse.py tests/test_scatter_range.py tests/test_dense_parity.py -m gpu -q -k "dense_linear or rangelib or half_conv_backward" > gpurun_out/h1_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/h1_tests.log
