// Shared by the 16-bit-MFMA fused-convolution code: conv_wave5h.hip (wave-autonomous row-block groups, ticket commit),
// conv_wave6h.hip (weight-stationary ranges) and weights_multi.hip. They read the same prepared weights (MFMA fragment order, pcs_conv_prepare_weights_h) and share the tile epilogue of conv_common.h.
#pragma once
#include "conv_common.h"

namespace pcs {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f32x4 mfma_h(Bf16, const uint4 &a, const uint4 &b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma_h(Fp16, const uint4 &a, const uint4 &b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// local column (inside a CT-wide column tile) that lane n of 16-column tile tl feeds -- the interleave the commit and
// the epilogue of the wave kernels assume (quads of 4 tiles: 64 q + 4 n + f; a pair: + 2 n + f; a single: + n)
__host__ __device__ inline int h_local_col(int nctt, int tl, int n) {
#if PCS_COMMIT_ATOMIC
  // atomic commit (one ds_add_f32 per lane and element): tile tl owns the 16 CONSECUTIVE columns 16 tl .. 16 tl + 15, so
  // the 16 lanes of a row hit 16 consecutive LDS banks (the 4-interleave put every lane on banks = f mod 4: 4-way
  // conflicts). The fragment order of the prepared weights makes any column assignment free on the operand side.
  (void)nctt;
  return 16 * tl + n;
#endif
  const int n4 = nctt / 4, n2 = (nctt % 4) / 2;
  if (tl < 4 * n4) return 64 * (tl / 4) + 4 * n + (tl % 4);
  if (tl < 4 * n4 + 2 * n2) return 64 * n4 + 2 * n + (tl - 4 * n4);
  return 64 * n4 + 32 * n2 + n;
}

struct ConvArgsH {
  const char *src;    // (n_src, cin) halfs
  const char *Wp;     // prepared weights, fragment order
  const float *bias;  // fp32, may be NULL
  uint16_t *dst;      // (n_dst, cout) halfs
  const int32_t *pairs;
  const int32_t *seg;
  int64_t n_dst;
  int64_t ntiles;
  int cin, cout, K, src_col, ncoltiles, xcd_remap, tile_rows, nt16, ns;
  double *stats;  // optional [ntiles][2][cout], as ConvArgs::stats (over the ROUNDED values stored)
  const int32_t *order;  // optional [ntiles]: workgroup slot -> row tile (heaviest first), as ConvArgs::order
  const uint16_t *addend = nullptr;  // optional (n_dst, cout) halfs: added (in fp32, before the rounding) to the output rows
  float act_slope = 1.f;              // LeakyReLU in the write-back, as ConvArgs::act_slope
  const void *gs_x = nullptr;         // BatchNorm backward statistics in the write-back, as ConvArgs::gs_* (x in the storage dtype)
  const uint32_t *gs_mask = nullptr;
  const double *gs_stat = nullptr;
};

template <typename HT> struct GsType;
template <> struct GsType<Bf16> { static constexpr int value = kGsBf16; };
template <> struct GsType<Fp16> { static constexpr int value = kGsFp16; };

}  // namespace pcs
