// Sparse 3-D convolution on gfx950: output-stationary fused gather-GEMM-accumulate for forward, dgrad and transposed
// convolutions, fp32 storage + fp32 MFMA (v_mfma_f32_16x16x4_f32). This file: the per-layer launch shape and the C
// entry point; the kernels live in conv_wave5.hip (>= 64 channels), conv_wave4.hip (other 16-byte-granular shapes)
// and conv_block.hip (everything else); the weight gradient in conv_wgrad.hip.
#include <math.h>

#include "conv_half.h"

using namespace pcs;

namespace pcs {
bool conv6h_applies(int cin, int cout, int K);       // conv_wave6h.hip
bool conv6h_is_chunked(int cin, int cout);
int conv6h_mode();
}

namespace {

int device_cus() {
  static int cus = 0;
  if (!cus) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) cus = n;
    else cus = 256;
  }
  return cus;
}

// dst[k][b][a] = src[k][a][b]: per-offset transposed weights for dgrad (a few MB per layer, once per backward pass).
// 32x32 LDS tiles (+1 padding), 128-byte lines on both sides -- the generic strided copy ran at ~0.7 TB/s here.
__global__ void __launch_bounds__(256) transpose_kab_kernel(const float *__restrict__ src, int A, int B,
                                                            float *__restrict__ dst) {
  __shared__ float tile[32][33];
  const int k = blockIdx.z, a0 = blockIdx.y * 32, b0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const float *s = src + (int64_t)k * A * B;
  float *d = dst + (int64_t)k * A * B;
#pragma unroll
  for (int r = ty; r < 32; r += 8)
    if (a0 + r < A && b0 + tx < B) tile[r][tx] = s[(int64_t)(a0 + r) * B + b0 + tx];
  __syncthreads();
#pragma unroll
  for (int r = ty; r < 32; r += 8)
    if (b0 + r < B && a0 + tx < A) d[(int64_t)(b0 + r) * A + a0 + tx] = tile[tx][r];
}

}  // namespace

extern "C" int pcs_transpose_kab_f32(const float *src, int32_t K, int32_t A, int32_t B, float *dst, void *stream) {
  if (K <= 0 || A <= 0 || B <= 0 || K > 65535 || !src || !dst) { set_error("pcs_transpose_kab_f32: bad args"); return PCS_EINVAL; }
  hipLaunchKernelGGL(transpose_kab_kernel, dim3((unsigned)ceil_div(B, 32), (unsigned)ceil_div(A, 32), (unsigned)K), dim3(256),
                     0, as_stream(stream), src, A, B, dst);
  return check_launch("pcs_transpose_kab_f32");
}

// Bumped whenever a fused-conv kernel, its launch shape picker or its epilogue changes: measurements keyed to kernels
// (profiles/*_conv_traffic.json) carry the revision they were taken on and bench.py refuses a stale one.
extern "C" const char *pcs_conv_kernel_revision(void) { return "r6.0"; }

extern "C" int32_t pcs_conv_tile_rows(int32_t cin, int32_t cout) {
  (void)cin;
  (void)cout;
  return 128;
}

// Output tile height for one layer call (profiles/round2_tile_sweep_{f32,bf16}.md: every k3 shape of MinkUNet-34 on the
// 12-frame maps x 13 heights). Model of one launch: the workgroups of a CU share its MFMA pipe, so the launch takes
// (workgroups per CU) x (work of one workgroup), work = pairs per tile + the padding of half a 16-row block per offset;
// the last, partly filled round of workgroups counts half (heaviest-first tile order: the light tiles fill it).
// Tall tiles pad less but fit fewer workgroups per CU (LDS holds the fp32 accumulator tile).
namespace {
double launch_cost(int64_t n_dst, int T, int64_t ncol, double ppr, int K) {
  const double per_cu = (double)(ceil_div(n_dst, T) * ncol) / (double)device_cus();
  const double rounds = 0.5 * (ceil(per_cu) + per_cu);
  return rounds * (T * ppr + 8.0 * K);
}
}  // namespace

extern "C" int32_t pcs_conv_pick_tile_rows_dt(int64_t n_dst, int64_t n_pairs, int32_t K, int32_t cin, int32_t cout,
                                              int32_t dtype) {
  static const int fixed = getenv("PCS_CONV_TILE") ? atoi(getenv("PCS_CONV_TILE")) : 0;
  if (n_dst <= 0 || K <= 0 || cin <= 0 || cout <= 0) return 128;
  if (!(dtype == 0 ? conv5_applies(cin, cout, K) : convh_applies(cin, cout, K))) return 128;
  if (fixed > 0) return fixed;
  const double ppr = (double)n_pairs / (double)n_dst;
  const int nctt = conv_nctt(cout);
  if (dtype != 0) {
    // 16-bit MFMA kernels: bound by the operand stream and, on the sparse levels, by the serial commit chain of a
    // workgroup -- two (or more) 4-wave workgroups per CU beat one tall 8-wave workgroup except on the >= 256-channel
    // deep levels, where the tall tile's lower padding and W reuse win (224 / 288 rows)
    if (cin >= 256 && cout >= 256) {
      // the weight-stationary kernel's chunked instances (conv_wave6h.hip) want two 4-wave workgroups per CU: 1.06-1.18x on 384
      // channels and on the small stride-16 level; 256 -> 256 on a big level stays on conv_os5h's tall 8-wave tile (0.97x)
      if (conv6h_mode() && conv6h_applies(cin, cout, K) && conv6h_is_chunked(cin, cout) && (cin > 256 || cout > 256 || n_dst <= 65536))
        return 144;
      const int64_t ncol = ceil_div(cout, 16 * nctt);
      return launch_cost(n_dst, 288, ncol, ppr, K) <= launch_cost(n_dst, 224, ncol, ppr, K) ? 288 : 224;
    }
    // up to the tallest tile that fits twice per CU (192 rows at 96 columns, 144 at 128); below a few rounds of
    // workgroups the height that fills the last round
    const int64_t ncol = ceil_div(cout, 16 * nctt);
    const int tmax = nctt == 6 ? 192 : 144;
    int best = tmax;
    double best_cost = launch_cost(n_dst, tmax, ncol, ppr, K);
    for (int T = tmax - 16; T >= tmax - 48; T -= 16) {
      const double c = launch_cost(n_dst, T, ncol, ppr, K);
      if (c < best_cost * 0.97) { best = T; best_cost = c; }
    }
    return best;
  }
  if (conv5_nctt(cout, 192) <= 4) {
    // <= 64-column tiles (32 / 64 outputs, or >= 128 outputs as 64-column tiles): 192..288 rows, 2-3 workgroups per CU
    const int64_t ncol = ceil_div(cout, 16 * conv5_nctt(cout, 192));
    int best = 224;
    double best_cost = 0;
    for (int T = 192; T <= 288; T += 32) {
      const double c = launch_cost(n_dst, T, ncol, ppr, K);
      if (best_cost == 0 || c < best_cost) { best = T; best_cost = c; }
    }
    return best;
  }
  // 96-column tiles (and other shapes of six / eight 16-column tiles that are not multiples of 64)
  const int64_t ncol = ceil_div(cout, 16 * nctt);
  const int64_t slots = (int64_t)device_cus() * 2;
  // many rounds of workgroups and few pairs per row (strides 1/2: 4-5.5 pairs per row, 1.28-1.36x MFMA padding at
  // 128 rows): one 8-wave workgroup per CU on the tallest tile the LDS holds pads 1.12-1.17x
  if (ceil_div(n_dst, 128) * ncol >= 8 * slots && ppr < 6.5) {
    int T = 384;
    while (T > 128 && conv5_lds_est(T, nctt) > kMaxDynLds) T -= 32;
    if (T >= 192) return T;
  }
  int best = 128;
  double best_cost = launch_cost(n_dst, 128, ncol, ppr, K) * 0.97;
  for (int T = 80; T <= 192; T += 16) {
    if (T == 128) continue;
    if (2 * conv5_lds_est(T, nctt) > 160 * 1024) continue;  // keep two workgroups per CU
    const double c = launch_cost(n_dst, T, ncol, ppr, K);
    if (c < best_cost) { best = T; best_cost = c; }
  }
  return best;
}

extern "C" int32_t pcs_conv_pick_tile_rows(int64_t n_dst, int64_t n_pairs, int32_t K, int32_t cin, int32_t cout) {
  return pcs_conv_pick_tile_rows_dt(n_dst, n_pairs, K, cin, cout, 0);
}

// 1 when the fused convolution of this shape can leave per-tile BatchNorm partial sums in its write-back (the wave
// kernels; the tile must hold the reduction scratch). dtype 0: fp32 kernels, 1 / 2: half kernels.
extern "C" int32_t pcs_conv_emits_bn_partials(int32_t cin, int32_t cout, int32_t K, int32_t tile_rows, int32_t dtype) {
  if (cin <= 0 || cout <= 0 || K <= 0 || K > 32 || (cin % 4) || (cout % 4) || tile_rows < 16) return 0;
  int nctt = conv_nctt(cout), nt = 256;
  if (dtype == 0) {
    if (conv5_applies(cin, cout, K)) {
      nctt = conv5_nctt(cout, tile_rows);
      nt = 2 * conv5_lds_est(tile_rows, nctt) > 160 * 1024 ? 512 : 256;
    } else if (tile_rows != 64 && tile_rows != 128) {
      return 0;
    }
  } else {
    if (!convh_applies(cin, cout, K)) return 0;
    nt = 2 * conv5_lds_est(tile_rows, nctt) > 160 * 1024 ? 512 : 256;
  }
  return conv_stats_fit(tile_rows, 16 * nctt, nt) ? 1 : 0;
}

// 1 when the fused convolution of this shape reads `tile_order` (the wave-autonomous kernels). dtype as above.
extern "C" int32_t pcs_conv_uses_tile_order(int32_t cin, int32_t cout, int32_t K, int32_t dtype) {
  if (cin <= 0 || cout <= 0 || K <= 0) return 0;
  return (dtype == 0 ? conv5_applies(cin, cout, K) : convh_applies(cin, cout, K)) ? 1 : 0;
}

extern "C" int32_t pcs_conv_supports_epilogue(int32_t cin, int32_t cout, int32_t K, int32_t dtype) {
  if (cin <= 0 || cout <= 0 || K <= 0 || K > 32) return 0;
  if (dtype == 0) return (cin % 4 == 0 && cout % 4 == 0) ? 1 : 0;   // the wave kernels (conv_os5 / conv_os4)
  return convh_applies(cin, cout, K) ? 1 : 0;
}

extern "C" int pcs_conv_gather_gemm_f32(const float *src, int64_t n_src, int32_t cin,
                                        const float *W, int32_t K, int32_t cout,
                                        const int32_t *pairs, int32_t src_col,
                                        const int32_t *seg, int32_t tile_rows, int64_t n_dst,
                                        const float *bias, float *dst, double *bn_partial,
                                        const int32_t *tile_order, void *stream) {
  return pcs_conv_gather_gemm_f32_ex(src, n_src, cin, W, K, cout, pairs, src_col, seg, tile_rows, n_dst, bias, nullptr, dst,
                                     bn_partial, tile_order, stream);
}

extern "C" int pcs_conv_gather_gemm_f32_ex(const float *src, int64_t n_src, int32_t cin,
                                           const float *W, int32_t K, int32_t cout,
                                           const int32_t *pairs, int32_t src_col,
                                           const int32_t *seg, int32_t tile_rows, int64_t n_dst,
                                           const float *bias, const pcs_conv_epilogue *ep, float *dst, double *bn_partial,
                                           const int32_t *tile_order, void *stream) {
  const float *addend = ep ? reinterpret_cast<const float *>(ep->addend) : nullptr;
  if (cin <= 0 || cout <= 0 || K <= 0 || n_dst < 0 || n_src < 0 || (src_col != 0 && src_col != 1)) {
    set_error("pcs_conv_gather_gemm_f32: bad sizes");
    return PCS_EINVAL;
  }
  if (n_dst == 0) return PCS_OK;
  if (!W || !seg || !dst || (n_src > 0 && !src)) { set_error("pcs_conv_gather_gemm_f32: null pointer"); return PCS_EINVAL; }
  if (tile_rows < 16 || tile_rows > 512 || tile_rows % 16) { set_error("pcs_conv_gather_gemm_f32: tile_rows must be a multiple of 16 in [16, 512]"); return PCS_EINVAL; }
  ConvArgs a;
  a.src = src; a.W = W; a.bias = bias; a.dst = dst; a.pairs = pairs; a.seg = seg;
  a.n_dst = n_dst; a.ntiles = ceil_div(n_dst, tile_rows); a.tile_rows = tile_rows;
  a.cin = cin; a.cout = cout; a.K = K; a.src_col = src_col; a.ncoltiles = 1; a.stats = bn_partial;
  a.order = tile_order;
  a.addend = addend;
  if (addend && ((uintptr_t)addend & 15)) { set_error("pcs_conv_gather_gemm_f32_ex: misaligned addend"); return PCS_EINVAL; }
  if (ep && ep->bn_x) {
    if (!bn_partial || !ep->bn_stat || ((uintptr_t)ep->bn_x & 15) || (ep->bn_mask && (cout & 31))) {
      set_error("pcs_conv_gather_gemm_f32_ex: BatchNorm backward statistics need bn_partial, bn_stat, 16-byte aligned bn_x and cout %% 32 == 0 with a gate mask");
      return PCS_EINVAL;
    }
    a.gs_x = ep->bn_x; a.gs_mask = ep->bn_mask; a.gs_stat = ep->bn_stat;
  }
  if (ep && ep->act_slope != 0.f && ep->act_slope != 1.f) a.act_slope = ep->act_slope;
  if (bn_partial && !pcs_conv_emits_bn_partials(cin, cout, K, tile_rows, 0)) {
    set_error("pcs_conv_gather_gemm_f32: this shape / tile height does not produce BatchNorm partials (ask pcs_conv_emits_bn_partials)");
    return PCS_EUNSUPPORTED;
  }
  static const int xcd = getenv("PCS_CONV_XCD") ? atoi(getenv("PCS_CONV_XCD")) : 1;  // 0: no XCD-contiguous tile order (debug)
  // 1: tiles in row order, one contiguous range per XCD; 2 (with a tile order): tiles dealt round-robin over the XCDs,
  // the column tiles of a row tile back to back on one XCD (they gather the same A rows: +0.5..1 %)
  a.xcd_remap = xcd ? (tile_order ? 2 : 1) : 0;
  const bool vec = (cin % 4 == 0) && (cout % 4 == 0) && (((uintptr_t)src | (uintptr_t)W | (uintptr_t)dst | (uintptr_t)bias) & 15) == 0;
  hipStream_t st = as_stream(stream);
  static const int generic = getenv("PCS_CONV_V1") ? atoi(getenv("PCS_CONV_V1")) : 0;  // 1: generic kernel only (debug)
  if (bn_partial && (!vec || generic || (!conv5_applies(cin, cout, K) && K > 32))) {
    set_error("pcs_conv_gather_gemm_f32: BatchNorm partials need the 16-byte-granular wave kernels");
    return PCS_EUNSUPPORTED;
  }
  if ((addend || a.gs_x || a.act_slope != 1.f) && !(vec && !generic && K <= 32)) {
    set_error("pcs_conv_gather_gemm_f32_ex: this shape runs on the generic kernel, which takes no write-back extras (pcs_conv_supports_epilogue)");
    return PCS_EUNSUPPORTED;
  }
  if (vec && !generic && conv5_applies(cin, cout, K)) return launch_conv_wave5(a, st);
  if (tile_rows != 64 && tile_rows != 128) { set_error("pcs_conv_gather_gemm_f32: this shape takes tile_rows 64 or 128"); return PCS_EUNSUPPORTED; }
  if (vec && !generic && K <= 32) return launch_conv_wave4(a, st);
  return launch_conv_block(a, vec, st);
}
