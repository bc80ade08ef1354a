mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/gputest.log 2>&1; tail -5 gpurun_out/gputest.log
SH="0 96 96 0|1 96 96 0|2 128 128 0|2 192 128 0|3 256 256 0|4 256 256 0|1 64 64 0|0 32 32 0"
SHH="0 96 96 0 bf16|1 96 96 0 bf16|2 128 128 0 bf16|3 256 256 0 bf16|4 256 256 0 bf16|1 64 64 0 bf16"
IFS='|' read -ra A <<< "$SH"; IFS='|' read -ra B <<< "$SHH"
for v in product rmw wait; do
  if [ $v = product ]; then unset PCS_LIB_PATH; else export PCS_LIB_PATH=$PWD/openpcseg_amd/lib/dbg/$v.so; fi
  echo "== $v" >> gpurun_out/ab_commit.txt
  PCS_SWEEP_REPS=60 timeout 400 python tools/conv_tile_sweep.py "${A[@]}" "${B[@]}" >> gpurun_out/ab_commit.txt 2>&1
done
unset PCS_LIB_PATH
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r3c1_bench.log 2> gpurun_out/r3c1_bench.err; tail -c 1500 gpurun_out/r3c1_bench.log
bash tools/convh_pmc.sh r3c1_convh_pmc
cat gpurun_out/r3c1_convh_pmc.err | tail -14
