cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_fuse.py tests/test_trajectory.py tests/test_ring_variant.py -m gpu -q > gpurun_out/b_tests1.log 2>&1; echo "tests1 rc=$?"; tail -5 gpurun_out/b_tests1.log
timeout 900 python -m pytest tests/test_fullsize_parity.py -m gpu -q -k "mk34 or minkunet34 or fused" > gpurun_out/b_tests2.log 2>&1; echo "tests2 rc=$?"; tail -4 gpurun_out/b_tests2.log
timeout 600 python tools/modelbench.py minkunet34:fuse,spvcnn18:reference,spvcnn18:fuse,rpvnet34:reference,rpvnet34:fuse > gpurun_out/b_modelbench.json 2> gpurun_out/b_modelbench.err; cat gpurun_out/b_modelbench.json
for s in minkunet34:fuse:bf16 minkunet34:workload:bf16 minkunet34:reference:bf16; do timeout 300 python tools/host_profile_model.py $s 45 > gpurun_out/b_host_$(echo $s | tr ':' '_').txt 2>&1; head -2 gpurun_out/b_host_$(echo $s | tr ':' '_').txt | tail -1; done
