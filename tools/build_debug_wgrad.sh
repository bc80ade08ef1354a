#!/bin/bash
# Debug variant of the library with conv_wgrad.hip recompiled: bash tools/build_debug_wgrad.sh <name> <-Dflags...>
#   e.g. tools/build_debug_wgrad.sh wab1 -DPCS_ABLATEW=1  -> openpcseg_amd/lib/dbg/<name>.so (select with PCS_LIB_PATH)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
python -m openpcseg_amd.build > /dev/null
mkdir -p $ROOT/openpcseg_amd/lib/dbg
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -munsafe-fp-atomics -fPIC -Wno-unused-value "$@" \
  -c $ROOT/openpcseg_amd/csrc/conv_wgrad.hip -o /tmp/conv_wgrad_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/conv_wgrad_$name.o \
  $(ls $ROOT/openpcseg_amd/lib/*.hip.o | grep -v conv_wgrad.hip.o) -o $ROOT/openpcseg_amd/lib/dbg/$name.so
echo $ROOT/openpcseg_amd/lib/dbg/$name.so
