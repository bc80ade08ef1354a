cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/round5_gputest.txt 2>&1; echo "gpu tests rc=$?"; tail -4 gpurun_out/round5_gputest.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
