#!/bin/bash
# Debug variant of the library with only the half-precision conv translation units recompiled:
#   bash tools/build_debug_convh.sh <name> <-Dflags...>      e.g. tools/build_debug_convh.sh hab2 -DPCS_ABLATEH=2
# -> openpcseg_amd/lib/dbg/<name>.so (conv_wave5h.hip and, when present, conv_wave6h.hip recompiled with the flags, the other
# objects of the product build linked as they are); select it with PCS_LIB_PATH. Debug builds are never loaded by default.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
python -m openpcseg_amd.build > /dev/null
mkdir -p $ROOT/openpcseg_amd/lib/dbg /tmp/pcsh_$name
rm -f /tmp/pcsh_$name/*.o
skip=""
for f in conv_wave5h.hip conv_wave6h.hip; do
  [ -f $ROOT/openpcseg_amd/csrc/$f ] || continue
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -munsafe-fp-atomics -fPIC -Wno-unused-value -Wno-array-bounds \
    -fno-slp-vectorize "$@" -c $ROOT/openpcseg_amd/csrc/$f -o /tmp/pcsh_$name/$f.o &
  skip="$skip\|/$f.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/pcsh_$name/*.o \
  $(ls $ROOT/openpcseg_amd/lib/*.hip.o | grep -v "/nonexistent$skip") -o $ROOT/openpcseg_amd/lib/dbg/$name.so
echo $ROOT/openpcseg_amd/lib/dbg/$name.so
