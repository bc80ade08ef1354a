#!/bin/bash
# A/B variant of the library: bash tools/build_variant_lib.sh <name> <-Dflags...>
#   e.g. tools/build_variant_lib.sh rmw -DPCS_COMMIT_ATOMIC=0 ; tools/build_variant_lib.sh wait -DPCS_COMMIT_NOWAIT=0
# -> openpcseg_amd/lib/dbg/<name>.so: every conv*.hip recompiled with the flags (conv_common.h switches reach the launch
# shape code too), the other objects of the product build linked as they are; select it with PCS_LIB_PATH.
# Variant builds are never loaded by default and are removed before a round ends (they must not travel as product).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
python -m openpcseg_amd.build > /dev/null
mkdir -p $ROOT/openpcseg_amd/lib/dbg /tmp/pcsvar_$name
rm -f /tmp/pcsvar_$name/*.o
CC="/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -munsafe-fp-atomics -fPIC -Wno-unused-value -Wno-array-bounds"
for f in $ROOT/openpcseg_amd/csrc/conv*.hip; do
  extra=""; case $(basename $f) in conv_wave5*.hip) extra="-fno-slp-vectorize";; esac  # as openpcseg_amd/build.py EXTRA_FLAGS
  $CC $extra "$@" -c $f -o /tmp/pcsvar_$name/$(basename $f).o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/pcsvar_$name/*.o \
  $(ls $ROOT/openpcseg_amd/lib/*.hip.o | grep -v "/conv") -o $ROOT/openpcseg_amd/lib/dbg/$name.so
echo $ROOT/openpcseg_amd/lib/dbg/$name.so
