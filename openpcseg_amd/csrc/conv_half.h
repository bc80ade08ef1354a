// Shared by the 16-bit-MFMA fused-convolution kernels: conv_wave5h.hip (wave-autonomous row-block groups, ticket commit)
// and conv_ring6h.hip (column-parallel waves, gathered rows through an LDS ring). Both read the same prepared weights
// (MFMA fragment order, pcs_conv_prepare_weights_h) and share the tile epilogue of conv_common.h.
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
  int ring_bt_cap, ring_acc_off;  // ring kernel: capacity of its batch table (entries), byte offset of the accumulator tile in LDS
};

// ---- the ring kernel's launch shape (conv_ring6h.hip), shared with the tile-height picker and the BatchNorm-partials query ----
#ifndef PCS_RING_D
#define PCS_RING_D 5             /* variant builds (tools/build_variant_lib.sh): ring depth */
#endif
#ifndef PCS_RING_ABLATE
#define PCS_RING_ABLATE 0        /* variant builds: 1 no weight loads, 2 row DMAs of 4 bytes per lane, 3 no commit, 4 no MFMA, 5 no row DMA at all,
                                    6 no DMA completion wait, 7 compute waves pass the barriers only (results are wrong, times tell) */
#endif
constexpr int kRingDepth = PCS_RING_D;  // A ring: batches of gathered rows resident / in flight per workgroup
constexpr int kRingBatchRows = 2;  // 16-row blocks per batch
constexpr int kRingMeta = PCS_RING_D > 9 ? 32 : 16;  // pair-index ring: slots (>= 2 kRingDepth - 2)

struct RingShape {
  int nctt, nc, kc;  // 16-column tiles per column tile, of them per compute wave, 32-channel steps per weight chunk
  int nwaves() const { return nctt / nc + 1; }  // compute waves + the loader
};

// the ring kernel serves cin % 32 == 0 from 64 channels with 2, 3 or 4 steps per chunk (cin = 64, 96, 128, 192, 256, 384, 512 ...)
// and 96 / 128-column tiles (cout = 96, 128, 192, 256, 384 ...)
int &conv_ring_mode();  // conv_ring6h.hip: 0 never, 1 wherever it applies, -1 per-shape policy (pcs_conv_ring_enable / PCS_CONVH_RING)
inline bool conv_ring_policy(int cin, int cout, int K) {
  // where the ring kernel beats conv_os5h_kernel (profiles/round4_ring.md); nowhere yet
  (void)cin; (void)cout; (void)K;
  return false;
}
inline bool conv_ring_shape(int cin, int cout, int K, RingShape *out) {
  static const int force_nc = getenv("PCS_CONVH_RING_NC") ? atoi(getenv("PCS_CONVH_RING_NC")) : 0;
  const int mode = conv_ring_mode();
  if (mode == 0 || !convh_applies(cin, cout, K) || cin % 32) return false;
  if (mode < 0 && !conv_ring_policy(cin, cout, K)) return false;
  const int ns = cin / 32, nctt = conv_nctt(cout);
  if (nctt != 6 && nctt != 8) return false;
  int kc = 0;
  if (ns % 4 == 0) kc = 4;
  else if (ns % 3 == 0) kc = 3;
  else if (ns % 2 == 0) kc = 2;
  if (!kc || ns / kc > 7) return false;
  if (out) { out->nctt = nctt; out->nc = (force_nc == 1 || force_nc == 2) ? force_nc : 2; out->kc = kc; }
  return true;
}
inline int conv_ring_bt_cap(int T, int ns, int kc, int K) {
  const int per_off = (T / 16 + kRingBatchRows - 1) / kRingBatchRows + 1;
  return ((ns / kc) * K * per_off + 15) & ~15;
}
// LDS bytes: [A ring | pair-index ring | offset lists | batch table | accumulator tile]; returns the tile's byte offset in *acc_off
inline size_t conv_ring_lds(int T, const RingShape &s, int ns, int K, int *acc_off) {
  size_t off = (size_t)kRingDepth * kRingBatchRows * s.kc * 1024 + (size_t)kRingMeta * kRingBatchRows * 128;
  off += 4 * 36 * 4 + 16;                                   // kl_k / kl_s / kl_m / kl_b + {nk, NB}
  off = (off + 7) & ~(size_t)7;
  off += (size_t)conv_ring_bt_cap(T, ns, s.kc, K) * 8;      // int2 descriptors
  off = (off + 15) & ~(size_t)15;
  if (acc_off) *acc_off = (int)off;
  return off + (size_t)(T + 1) * (16 * s.nctt + 4) * 4;
}
inline bool conv_ring_applies(int cin, int cout, int K, int T, RingShape *out) {
  RingShape s;
  if (!conv_ring_shape(cin, cout, K, &s)) return false;
  if (T < 32 || T > 512 || conv_ring_lds(T, s, cin / 32, K, nullptr) > kMaxDynLds) return false;
  if (out) *out = s;
  return true;
}
// tallest tile the ring kernel's LDS layout holds
inline int conv_ring_max_rows(int cin, int cout, int K) {
  RingShape s;
  if (!conv_ring_shape(cin, cout, K, &s)) return 0;
  int T = 512;
  while (T >= 32 && conv_ring_lds(T, s, cin / 32, K, nullptr) > kMaxDynLds) T -= 16;
  return T >= 32 ? T : 0;
}

int launch_conv_ring6h(const ConvArgsH &a, int dtype, hipStream_t st);  // conv_ring_applies(); dtype 1 bf16, 2 fp16

}  // namespace pcs
