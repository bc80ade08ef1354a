mkdir -p gpurun_out; : > gpurun_out/r3c5_convh_R.txt
for r in 2 3 4; do
  echo "== PCS_CONVH_R=$r" >> gpurun_out/r3c5_convh_R.txt
  PCS_CONVH_R=$r PCS_SWEEP_REPS=60 timeout 300 python tools/conv_tile_sweep.py "0 96 96 0 bf16" "1 96 96 0 bf16" "0 128 96 0 bf16" "1 64 64 0 bf16" "2 64 64 0 bf16" "2 128 64 0 bf16" "2 128 128 0 bf16" "3 256 256 0 bf16" "4 256 256 0 bf16" >> gpurun_out/r3c5_convh_R.txt 2>&1
done
grep -v amdgpu gpurun_out/r3c5_convh_R.txt | cut -c1-110
