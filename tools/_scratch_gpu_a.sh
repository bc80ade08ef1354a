# round 5, GPU call A: the new tests + the ref / ref+fuse / fused throughput of MinkUNet-34 + step budgets of the ref+fuse route
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_fuse.py tests/test_trajectory.py tests/test_ring_variant.py -m gpu -x -q > gpurun_out/a_tests1.log 2>&1; echo "tests1 rc=$?"; tail -3 gpurun_out/a_tests1.log
timeout 900 python -m pytest tests/test_fullsize_parity.py -m gpu -q -k "mk34 or fused" > gpurun_out/a_tests2.log 2>&1; echo "tests2 rc=$?"; tail -3 gpurun_out/a_tests2.log
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -q -k "weight_prep or weights_multi or fused_batchnorm or bn_" > gpurun_out/a_tests3.log 2>&1; echo "tests3 rc=$?"; tail -3 gpurun_out/a_tests3.log
timeout 600 python tools/modelbench.py minkunet34:reference,minkunet34:fuse,minkunet34:workload > gpurun_out/a_modelbench_mk34.json 2> gpurun_out/a_modelbench_mk34.err; cat gpurun_out/a_modelbench_mk34.json
timeout 500 bash tools/profile_model.sh a_mk34_fuse_f32 minkunet34:fuse:f32 > gpurun_out/a_prof_f32.log 2>&1; tail -22 gpurun_out/a_prof_f32.log
timeout 500 bash tools/profile_model.sh a_mk34_fuse_bf16 minkunet34:fuse:bf16 > gpurun_out/a_prof_bf16.log 2>&1; tail -22 gpurun_out/a_prof_bf16.log
