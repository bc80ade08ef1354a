#!/bin/bash
# Debug variant of the library: bash tools/build_debug_lib.sh <name> <-Dflags...>
#   e.g. tools/build_debug_lib.sh trace -DPCS_TRACE=1 ; tools/build_debug_lib.sh ab3 -DPCS_ABLATE5=3
# -> openpcseg_amd/lib/dbg/<name>.so (conv_wave5.hip recompiled with the flags, the other objects of the product
# build linked as they are); select it with PCS_LIB_PATH. Debug builds are never loaded by default.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
python -m openpcseg_amd.build > /dev/null
mkdir -p $ROOT/openpcseg_amd/lib/dbg
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -munsafe-fp-atomics -fPIC -Wno-unused-value "$@" \
  -c $ROOT/openpcseg_amd/csrc/conv_wave5.hip -o /tmp/conv_wave5_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/conv_wave5_$name.o \
  $(ls $ROOT/openpcseg_amd/lib/*.hip.o | grep -v conv_wave5.hip.o) -o $ROOT/openpcseg_amd/lib/dbg/$name.so
echo $ROOT/openpcseg_amd/lib/dbg/$name.so
