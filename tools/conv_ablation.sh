# A/B of the wave5 conv kernel's debug builds (-DPCS_ABLATE5=N, built here by tools/build_debug_lib.sh); see profiles/round1_conv_pmc.md
for n in 2 3 5 6; do bash tools/build_debug_lib.sh ab$n -DPCS_ABLATE5=$n > /dev/null; done
for shape in "3 256 256" "0 96 96" "2 128 128"; do
  for lib in "" ab2 ab3 ab5 ab6; do
    printf "%-12s %-4s " "$shape" "${lib:-full}"
    PCS_LIB_PATH=${lib:+$PWD/openpcseg_amd/lib/dbg/$lib.so} timeout 100 python tools/conv_microbench.py $shape 10 2>&1 | grep "^gemm" | sed 's/.*tile=None: //'
  done
done
