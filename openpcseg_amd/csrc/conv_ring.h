// Switch between the product build and the experimental "ring" variant.
// The column-parallel ring convolution kernels of round 4 (conv_ring6h / conv_ring6f: parity-green, 0.47-0.70x the wave kernels,
// profiles/round4_ring.md) are NOT in libpcseg_hip.so any more [r5]: their sources live in tools/experimental/csrc/ and are
// compiled only by `tools/build_variant_lib.sh ring` (-DPCS_WITH_RING=1), which also exports pcs_conv_ring_enable /
// pcs_conv_ring_applies. In the product build every ring query below is a constant "no", so the launch-shape code keeps one form.
#pragma once
#include "conv_half.h"

#ifndef PCS_WITH_RING
#define PCS_WITH_RING 0
#endif

#if PCS_WITH_RING
#include "conv_ring_impl.h"   // tools/experimental/csrc (the variant build adds the include path)
#else
namespace pcs {
struct RingShape {
  int nctt, nc, kc;
  int nwaves() const { return 1; }
};
inline bool conv_ring_applies(int, int, int, int, RingShape *) { return false; }
inline int conv_ring_max_rows(int, int, int) { return 0; }
inline bool conv_ringf_applies(int, int, int, int, RingShape *) { return false; }
inline int conv_ringf_max_rows(int, int, int) { return 0; }
inline int conv_ringf_nctt(int) { return 0; }
inline int launch_conv_ring6h(const ConvArgsH &, int, hipStream_t) { return PCS_EUNSUPPORTED; }
inline int launch_conv_ring6f(const ConvArgs &, hipStream_t) { return PCS_EUNSUPPORTED; }
}  // namespace pcs
#endif
