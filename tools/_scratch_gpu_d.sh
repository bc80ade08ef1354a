cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_range_sample.py tests/test_fuse.py tests/test_trajectory.py -m gpu -q > gpurun_out/d_tests1.log 2>&1; echo "tests1 rc=$?"; tail -6 gpurun_out/d_tests1.log
timeout 900 python -m pytest tests/test_fullsize_parity.py -m gpu -q -k "fused" > gpurun_out/d_tests2.log 2>&1; echo "tests2 rc=$?"; tail -4 gpurun_out/d_tests2.log
timeout 600 python tools/modelbench.py spvcnn18:fuse,rpvnet34:fuse > gpurun_out/d_modelbench.json 2> gpurun_out/d_modelbench.err; cat gpurun_out/d_modelbench.json
timeout 600 bash tools/profile_model.sh d_rpvnet34_fuse_f32 rpvnet34:fuse:f32 > gpurun_out/d_prof_rpv.log 2>&1; head -12 gpurun_out/d_rpvnet34_fuse_f32_step_budget.md | cut -c1-220
