mkdir -p gpurun_out
timeout 300 python tools/module_trace.py hip config4 40000 tools/_scratch/cfg4_ref.npz > gpurun_out/r3c4_cfg4_modules.txt 2>&1
: > gpurun_out/r3c4_trace_half.txt
for sh in "0 96 96" "3 256 256"; do
  echo "== trace $sh bf16" >> gpurun_out/r3c4_trace_half.txt
  PCS_LIB_PATH=$PWD/openpcseg_amd/lib/dbg/trace.so timeout 300 python tools/conv_trace.py $sh 12 bf16 >> gpurun_out/r3c4_trace_half.txt 2>&1
done
bash tools/convh_pmc.sh r3c4_convh_pmc
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r3c4_bench.log 2> gpurun_out/r3c4_bench.err
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-amp-line --wgrad fp32 > gpurun_out/r3c4_bench_wgrad_fp32.log 2>> gpurun_out/r3c4_bench.err
grep -o '"value": [0-9.]*' gpurun_out/r3c4_bench.log gpurun_out/r3c4_bench_wgrad_fp32.log
