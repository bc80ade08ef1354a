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
#ifndef PCS_ABLATEH
#define PCS_ABLATEH 0  /* wave5h, timing only (results are wrong): 1 weight fragments loaded at a group's first step only, 2 no weight
                          loads, 3 no ticket / commit, 4 = 2 + 3, 5 no A loads, 6 weight fragments read out of LDS without a fill */
#endif
#ifndef PCS_ALIAS
#define PCS_ALIAS 0    /* wave5: 1 every offset reads W[0], 2 A rows read sequentially instead of gathered */
#endif
#ifndef PCS_COMMIT_ATOMIC
#define PCS_COMMIT_ATOMIC 0  /* wave5 / wave5h: 1 = commit with ds_add_f32 in ticket order. Measured in round 3 and left off: the LDS
                                float atomic runs at ~1 lane per 4 clocks on gfx950 -- every shape 2-3x slower, fp32 and bf16 alike
                                (profiles/round3_commit_ab.txt). 0 = read-add-write under the ticket. */
#endif
#ifndef PCS_COMMIT_NOWAIT
#define PCS_COMMIT_NOWAIT 1  /* hand the ticket on with a bare ds_write_b32 behind the tile writes instead of waiting for their completion
                                (the LDS executes one wave's instructions in order; the next owner reads only after it has read the ticket) */
#endif
#ifndef PCS_COMMIT_PHASED
#define PCS_COMMIT_PHASED 1  /* read-add-write commit as three fenced phases (all reads, all adds, all writes) with the row addresses
                                formed before the ticket wait; 0 = the compiler's interleaving (round 2) */
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
  double *stats;  // optional [ntiles][2][cout]: per-tile column sums / sums of squares of the rows written (BatchNorm)
  const int32_t *order;  // optional [ntiles]: workgroup slot -> row tile (heaviest first), nullptr = row order
  const float *addend = nullptr;  // optional (n_dst, cout): added to the output rows in the write-back (the wave kernels only):
                                  // the skip gradient of a residual block riding in the dgrad of its first convolution
  // optional: this launch is a dgrad whose output IS the gradient dy of a BatchNorm (+ ReLU) output -- its write-back then leaves
  // that BatchNorm's backward statistics sum(g), sum(g xhat), g = dy [y > 0], per tile in `stats` (instead of the forward sums)
  float act_slope = 1.f;              // LeakyReLU in the write-back: v < 0 -> v * act_slope (1 = none), before the store and the statistics
  const void *gs_x = nullptr;         // the BatchNorm's input rows (n_dst, cout), dst's dtype
  const uint32_t *gs_mask = nullptr;  // its ReLU gate, one bit per element (n_dst, cout / 32 words), NULL = no ReLU
  const double *gs_stat = nullptr;    // mean[cout] | invstd[cout]
};

// element type of gs_x in the shared epilogues
enum { kGsF32 = 0, kGsBf16 = 1, kGsFp16 = 2 };
struct GStat {
  const void *x;
  const uint32_t *mask;
  const double *stat;
  int dtype;
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

// LDS accumulate of the wave kernels' commit: one ds_add_f32 per lane (no return value, nothing to wait for)
__device__ __forceinline__ void lds_add(float *p, float v) {
  (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

constexpr size_t kMaxDynLds = 160 * 1024 - 256;  // per-workgroup LDS ceiling of a gfx950 CU, minus the static part
// sink rows below the accumulator tile of the wave5 kernels (atomic commit: one per lane group) and the LDS estimate
// every launch-shape decision shares (tile + offset lists + slack)
constexpr int kConvSinkRows = PCS_COMMIT_ATOMIC ? 4 : 1;
inline size_t conv5_lds_est(int tile_rows, int nctt) { return (size_t)((tile_rows + kConvSinkRows) * (16 * nctt + 4)) * 4 + 1024; }

// 16-column MFMA tiles per column tile of the wave kernels: 1, 2, 3, 4, 6 or 8. Up to 128 output columns are one
// column tile; wider outputs take the width in {8, 6, 4} that pads the fewest columns (ties: the widest) -- 256 -> 8,
// 192 -> 6, and the cr 1.75 widths 168 -> 6 (192), 224 -> 8 (256), 336 -> 8 (384), 448 -> 4 (448), 672 -> 6 (672).
// The fragment order of the prepared half weights is defined by this function (conv_wave5h.hip).
inline int conv_nctt(int cout) {
  int nctt = (cout + 15) / 16;
  if (nctt <= 8) {
    if (nctt == 5) nctt = 6;
    if (nctt == 7) nctt = 8;
    return nctt;
  }
  int best = 8, best_pad = ((cout + 127) / 128) * 128 - cout;
  for (int n = 6; n >= 4; n -= 2) {
    const int pad = ((cout + 16 * n - 1) / (16 * n)) * 16 * n - cout;
    if (pad < best_pad) { best = n; best_pad = pad; }
  }
  return best;
}
// fp32 wave5: 16-column MFMA tiles per column tile for (cout, tile height). >= 128 output columns on tiles of 176 rows
// or more run as 64-column tiles (144 VGPRs: three waves per SIMD, two to three workgroups per CU; the taller tile pads
// fewer MFMA rows and the A rows read once per column tile cost nothing, profiles/round1_conv_pmc.md).
inline int conv5_nctt(int cout, int tile_rows) {
  if (cout >= 128 && cout % 64 == 0 && tile_rows >= 176) return 4;
  if (cout >= 192 && tile_rows >= 192) return 4;
  return conv_nctt(cout);
}
// wave5 serves every 16-byte-granular shape from 32 input channels up with an even column tile: cin % 32 == 0 on the
// straight-line pipeline, other cin % 4 == 0 on its TAIL instance (round 3: the cr 1.75 / cr 0.5 widths 56, 112, 168,
// 336, 48 ... used to fall to the one-row-block-per-step wave4 kernel)
inline bool conv5_applies(int cin, int cout, int K) {
  return cin % 4 == 0 && cin >= 32 && cout % 4 == 0 && K <= 32 && conv_nctt(cout) % 2 == 0;
}

// the half-precision wave kernel (conv_wave5h.hip) steps 32 channels per MFMA: rows are 16-byte granular from
// cin % 8 == 0 on; a last step of 8 / 16 / 24 channels meets zero-padded weight fragments (TAIL instance)
inline bool convh_applies(int cin, int cout, int K) {
  return cin % 8 == 0 && cin >= 32 && cout % 4 == 0 && K <= 32 && conv_nctt(cout) % 2 == 0;
}

// Tile epilogue of the wave kernels. The fp32 accumulator tile acc_l[T+1][ACS] is complete (the caller has passed its
// __syncthreads()). A thread owns one 4-column quad and every NRG-th row: each row is stored once (+bias) through
// `store(r, cq, v)`, which returns the value AS STORED (the rounded one for half outputs). With `stats` the kernel also
// leaves, per tile and column, sum(x) and sum(x^2) over the rows it wrote -- the statistics pass of the BatchNorm that
// follows the convolution: the sums are taken about the tile's first row in fp32, reduced over
// the row groups in a fixed order through the (now free) tile, and un-shifted in double (deterministic; needs
// T >= 2 NRG rows of scratch, conv_stats_fit()).
// four consecutive elements of a row-major (fp32 / bf16 / fp16) tensor, widened
__device__ __forceinline__ float4 gs_load4(const void *p, int dtype, int64_t idx) {
  if (dtype == kGsF32) return *reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(p) + idx);
  const uint2 u = *reinterpret_cast<const uint2 *>(reinterpret_cast<const uint16_t *>(p) + idx);
  if (dtype == kGsBf16)
    return make_float4(h2f(Bf16{}, (uint16_t)(u.x & 0xFFFFu)), h2f(Bf16{}, (uint16_t)(u.x >> 16)), h2f(Bf16{}, (uint16_t)(u.y & 0xFFFFu)),
                       h2f(Bf16{}, (uint16_t)(u.y >> 16)));
  return make_float4(h2f(Fp16{}, (uint16_t)(u.x & 0xFFFFu)), h2f(Fp16{}, (uint16_t)(u.x >> 16)), h2f(Fp16{}, (uint16_t)(u.y & 0xFFFFu)),
                     h2f(Fp16{}, (uint16_t)(u.y >> 16)));
}

// `gs` (with `stats`): BatchNorm BACKWARD statistics of the rows written -- sum(g), sum(g xhat) with g = stored value x ReLU gate,
// xhat = (x - mean) invstd of the BatchNorm whose output gradient this launch produces (ConvArgs::gs_*); row0 = the tile's first row.
template <int CT, int NT, typename Store>
__device__ __forceinline__ void conv_tile_epilogue(float *acc_l, int ACS, int rows, int n0, int cout, const float *bias,
                                                   double *stats, int tid, Store store, const GStat *gs = nullptr, int64_t row0 = 0) {
  constexpr int Q = CT / 4, NRG = NT / Q;
  const int q = tid % Q, rg = tid / Q, cq = 4 * q;
  const bool on = rg < NRG && n0 + cq < cout;
  float4 b = make_float4(0.f, 0.f, 0.f, 0.f), piv = b, s0 = b, s1 = b, gm = b, gi = b;
  if (on) {
    if (bias) b = *reinterpret_cast<const float4 *>(bias + n0 + cq);
    if (stats && !gs) {
      piv = *reinterpret_cast<const float4 *>(acc_l + cq);
      piv.x += b.x; piv.y += b.y; piv.z += b.z; piv.w += b.w;
    }
    if (stats && gs) {
      const int ch = n0 + cq;
      gm = make_float4((float)gs->stat[ch], (float)gs->stat[ch + 1], (float)gs->stat[ch + 2], (float)gs->stat[ch + 3]);
      gi = make_float4((float)gs->stat[cout + ch], (float)gs->stat[cout + ch + 1], (float)gs->stat[cout + ch + 2], (float)gs->stat[cout + ch + 3]);
    }
    for (int r = rg; r < rows; r += NRG) {
      float4 v = *reinterpret_cast<const float4 *>(acc_l + r * ACS + cq);
      v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
      const float4 sv = store(r, cq, v);
      if (stats && gs) {
        const int ch = n0 + cq;
        const float4 xv = gs_load4(gs->x, gs->dtype, (row0 + r) * (int64_t)cout + ch);
        unsigned bits = 0xFu;
        if (gs->mask) bits = (gs->mask[(row0 + r) * (int64_t)(cout >> 5) + (ch >> 5)] >> (ch & 31)) & 0xFu;
        const float g0 = (bits & 1u) ? sv.x : 0.f, g1 = (bits & 2u) ? sv.y : 0.f, g2 = (bits & 4u) ? sv.z : 0.f, g3 = (bits & 8u) ? sv.w : 0.f;
        s0.x += g0; s0.y += g1; s0.z += g2; s0.w += g3;
        s1.x += g0 * ((xv.x - gm.x) * gi.x); s1.y += g1 * ((xv.y - gm.y) * gi.y);
        s1.z += g2 * ((xv.z - gm.z) * gi.z); s1.w += g3 * ((xv.w - gm.w) * gi.w);
      } else if (stats) {
        const float dx = sv.x - piv.x, dy = sv.y - piv.y, dz = sv.z - piv.z, dw = sv.w - piv.w;
        s0.x += dx; s0.y += dy; s0.z += dz; s0.w += dw;
        s1.x += dx * dx; s1.y += dy * dy; s1.z += dz * dz; s1.w += dw * dw;
      }
    }
  }
  if (stats) {  // kernel argument: uniform over the launch
    __syncthreads();
    if (on) {
      *reinterpret_cast<float4 *>(acc_l + (2 * rg) * ACS + cq) = s0;
      *reinterpret_cast<float4 *>(acc_l + (2 * rg + 1) * ACS + cq) = s1;
      if (rg == 0) *reinterpret_cast<float4 *>(acc_l + (2 * NRG) * ACS + cq) = piv;
    }
    __syncthreads();
    if (tid < CT && n0 + tid < cout) {
      double t0 = 0.0, t1 = 0.0;
      for (int h = 0; h < NRG; ++h) { t0 += (double)acc_l[(2 * h) * ACS + tid]; t1 += (double)acc_l[(2 * h + 1) * ACS + tid]; }
      const double p = (double)acc_l[(2 * NRG) * ACS + tid], dn = (double)rows;
      stats[n0 + tid] = t0 + dn * p;
      stats[cout + n0 + tid] = t1 + 2.0 * p * t0 + dn * p * p;
    }
  }
}
// rows of tile scratch the statistics need: 2 per row group + the pivot row
inline bool conv_stats_fit(int tile_rows, int ct, int nt) { return tile_rows + 1 >= 2 * (nt / (ct / 4)) + 1; }

// launchers (each picks its template instance from the shape; `a.ncoltiles` is set inside)
int launch_conv_block(ConvArgs a, bool vec, hipStream_t st);   // any shape; tile_rows 64 / 128
int launch_conv_wave4(ConvArgs a, hipStream_t st);              // cin % 4 == 0, cout % 4 == 0, K <= 32; tile_rows 64 / 128
int launch_conv_wave5(ConvArgs a, hipStream_t st);              // conv5_applies(); any tile_rows % 16 == 0

}  // namespace pcs
