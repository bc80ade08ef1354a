cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/round5_bench.log 2> gpurun_out/round5_bench.err; echo "bench rc=$?"; tail -1 gpurun_out/round5_bench.log | wc -c; tail -1 gpurun_out/round5_bench.log; tail -4 gpurun_out/round5_bench.err
