mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/gputest.log 2>&1; tail -5 gpurun_out/gputest.log
SH="0 96 96 0|1 96 96 0|2 128 128 0|2 192 128 0|3 256 256 0|4 256 256 0|1 64 64 0|0 32 32 0|0 56 56 0|1 112 112 0|3 672 448 0"
SHH="0 96 96 0 bf16|1 96 96 0 bf16|2 128 128 0 bf16|3 256 256 0 bf16|4 256 256 0 bf16|1 64 64 0 bf16|0 56 56 0 bf16|1 112 112 0 bf16|3 672 448 0 bf16"
IFS='|' read -ra A <<< "$SH"; IFS='|' read -ra B <<< "$SHH"
: > gpurun_out/ab_commit2.txt
for v in product wait r2; do
  if [ $v = product ]; then unset PCS_LIB_PATH; else export PCS_LIB_PATH=$PWD/openpcseg_amd/lib/dbg/$v.so; fi
  echo "== $v" >> gpurun_out/ab_commit2.txt
  PCS_SWEEP_REPS=60 timeout 400 python tools/conv_tile_sweep.py "${A[@]}" "${B[@]}" >> gpurun_out/ab_commit2.txt 2>&1
done
unset PCS_LIB_PATH
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r3c2_bench.log 2> gpurun_out/r3c2_bench.err; tail -c 600 gpurun_out/r3c2_bench.log
