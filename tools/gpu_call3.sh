mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_fullsize_parity.py -m gpu -q > gpurun_out/gputest_fullsize.log 2>&1; tail -8 gpurun_out/gputest_fullsize.log
for sh in "0 96 96" "3 256 256" "2 128 128"; do
  echo "== trace $sh" >> gpurun_out/r3c3_trace.txt
  PCS_LIB_PATH=$PWD/openpcseg_amd/lib/dbg/trace.so timeout 300 python tools/conv_trace.py $sh >> gpurun_out/r3c3_trace.txt 2>&1
done
PCS_SWEEP_REPS=60 timeout 400 python tools/conv_tile_sweep.py "0 96 96 192,256,320,384" "1 96 96 192,256,320,384" "2 128 128 224,256,288" "3 256 256 224,256,288" "0 96 96 128,192,256,384 bf16" "3 256 256 144,224,288 bf16" > gpurun_out/r3c3_tiles.txt 2>&1
timeout 300 python tools/conv_layer_table.py 4 1.75 off > gpurun_out/r3c3_layer_table_cr175_fp32.txt 2>&1
timeout 300 python tools/conv_layer_table.py 4 1.75 bf16 > gpurun_out/r3c3_layer_table_cr175_bf16.txt 2>&1
tail -3 gpurun_out/r3c3_layer_table_cr175_fp32.txt gpurun_out/r3c3_layer_table_cr175_bf16.txt
