#!/bin/bash
# Where the time of the half fused convolution goes: timing-only ablation builds of conv_wave5h.hip (-DPCS_ABLATEH=N, results
# are wrong by construction) on the bench maps. Build first (CPU container): for n in 1 2 3 4 5 6; do bash
# tools/build_debug_convh.sh hab$n -DPCS_ABLATEH=$n; done      Output: gpurun_out/convh_ablation.txt
mkdir -p gpurun_out
out=gpurun_out/convh_ablation.txt
: > $out
export PCS_SWEEP_REPS=${PCS_SWEEP_REPS:-40}
for lib in "" hab1 hab2 hab3 hab4 hab5 hab6; do
  echo "== ${lib:-product}" >> $out
  PCS_LIB_PATH=${lib:+$PWD/openpcseg_amd/lib/dbg/$lib.so} timeout 300 python tools/conv_tile_sweep.py \
    "0 96 96 0,384 bf16" "1 96 96 0,384 bf16" "2 64 64 0 bf16" "2 128 128 0,224 bf16" "3 256 256 0 bf16" "4 256 256 0 bf16" 2>&1 | grep "^level" >> $out
done
cat $out
