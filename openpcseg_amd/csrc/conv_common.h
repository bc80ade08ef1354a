// Shared by the fused-convolution translation units (conv.hip = picker + C entry point, conv_block.hip = generic
// block-synchronous kernel, conv_wave4.hip / conv_wave5.hip = wave-autonomous kernels, conv_wgrad.hip = weight gradient).
#pragma once
#include <stdlib.h>

#include <type_traits>

#include "pcs_common.h"

// debug-build switches (tools/conv_ablation.sh, tools/conv_trace.py through PCS_LIB_PATH); all 0 in the product
#ifndef PCS_TRACE
#define PCS_TRACE 0    /* wave5: per-wave phase timers (wall clock, 10 ns) into g_conv_trace */
#endif
#ifndef PCS_ABLATE5
#define PCS_ABLATE5 0  /* wave5: 2 no MFMA, 3 no operand loads in the channel loop, 5 no W loads, 6 no A loads */
#endif
#ifndef PCS_ALIAS
#define PCS_ALIAS 0    /* wave5: 1 every offset reads W[0], 2 A rows read sequentially instead of gathered */
#endif
#if PCS_TRACE
#define PCS_T(...) __VA_ARGS__
#else
#define PCS_T(...)
#endif

namespace pcs {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct ConvArgs {
  const float *src;
  const float *W;
  const float *bias;
  float *dst;
  const int32_t *pairs;
  const int32_t *seg;
  int64_t n_dst;
  int64_t ntiles;
  int cin, cout, K, src_col, ncoltiles, xcd_remap, tile_rows;
};

// storage-format tags of the half-precision kernels (features / prepared weights / outputs)
struct Bf16 {};
struct Fp16 {};
struct Fp32 {};
__device__ __forceinline__ float h2f(Bf16, uint16_t h) { return __uint_as_float((uint32_t)h << 16); }
__device__ __forceinline__ float h2f(Fp16, uint16_t h) { return (float)__builtin_bit_cast(_Float16, h); }
__device__ __forceinline__ uint16_t f2h(Bf16, float f) {  // round to nearest even; NaN stays NaN
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (uint16_t)((u >> 16) | 0x40u);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
__device__ __forceinline__ uint16_t f2h(Fp16, float f) {
  const _Float16 h = (_Float16)f;
  return __builtin_bit_cast(uint16_t, h);
}

constexpr size_t kMaxDynLds = 160 * 1024 - 256;  // per-workgroup LDS ceiling of a gfx950 CU, minus the static part

// 16-column MFMA tiles per column tile of the wave kernels: 1, 2, 3, 4, 6 or 8
inline int conv_nctt(int cout) {
  int nctt = (cout + 15) / 16;
  if (nctt > 8) nctt = 8;
  if (nctt == 5) nctt = 6;
  if (nctt == 7) nctt = 8;
  return nctt;
}
// wave5 serves 16-byte-granular shapes with a contraction of at least two 32-channel steps and an even column tile
inline bool conv5_applies(int cin, int cout, int K) {
  return cin % 32 == 0 && cin >= 64 && cout % 4 == 0 && K <= 32 && conv_nctt(cout) % 2 == 0;
}

// the half-precision wave kernel (conv_wave5h.hip) steps 32 channels per MFMA and needs no second pipeline stage:
// every 16-byte-granular shape with a 32-channel contraction granule and an even column tile
inline bool convh_applies(int cin, int cout, int K) {
  return cin % 32 == 0 && cin >= 32 && cout % 4 == 0 && K <= 32 && conv_nctt(cout) % 2 == 0;
}

// launchers (each picks its template instance from the shape; `a.ncoltiles` is set inside)
int launch_conv_block(ConvArgs a, bool vec, hipStream_t st);   // any shape; tile_rows 64 / 128
int launch_conv_wave4(ConvArgs a, hipStream_t st);              // cin % 4 == 0, cout % 4 == 0, K <= 32; tile_rows 64 / 128
int launch_conv_wave5(ConvArgs a, hipStream_t st);              // conv5_applies(); any tile_rows % 16 == 0

}  // namespace pcs
