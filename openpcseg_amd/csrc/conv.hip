// Sparse 3-D convolution on gfx950: output-stationary fused gather-GEMM-accumulate for forward, dgrad and transposed
// convolutions, fp32 storage + fp32 MFMA (v_mfma_f32_16x16x4_f32). This file: the per-layer launch shape and the C
// entry point; the kernels live in conv_wave5.hip (>= 64 channels), conv_wave4.hip (other 16-byte-granular shapes)
// and conv_block.hip (everything else); the weight gradient in conv_wgrad.hip.
#include "conv_common.h"

using namespace pcs;

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
extern "C" const char *pcs_conv_kernel_revision(void) { return "r2.3-wave5+epilogue-stats"; }

extern "C" int32_t pcs_conv_tile_rows(int32_t cin, int32_t cout) {
  (void)cin;
  (void)cout;
  return 128;
}

// Output tile height for one layer. The workgroups of a launch run in waves of (CUs x 2) -- two 4-wave
// workgroups fit a CU -- so a launch of 1.1 waves takes as long as one of 2.0: with few output rows (strides
// 8/16) a height is chosen that fills the last wave (s16 256->256: 66.9 -> 82.7 TFLOP/s). Model: waves(T) x (pairs per tile + padding of half a
// 16-row block per offset); 128 unless another height is predicted >= 5 % faster.
extern "C" int32_t pcs_conv_pick_tile_rows(int64_t n_dst, int64_t n_pairs, int32_t K, int32_t cin, int32_t cout) {
  static const int fixed = getenv("PCS_CONV_TILE") ? atoi(getenv("PCS_CONV_TILE")) : 0;
  if (n_dst <= 0 || K <= 0 || !conv5_applies(cin, cout, K)) return 128;
  if (fixed > 0) return fixed;
  const int64_t slots = (int64_t)device_cus() * 2;
  const double ppr = (double)n_pairs / (double)n_dst;
  if (cout >= 192) {
    // wide outputs: 64-column tiles on 192..288-row tiles (two 4-wave workgroups per CU), see launch_conv_wave5
    const int64_t ncol64 = ceil_div(cout, 64);
    int best = 256;
    double best_cost = 0;
    for (int T = 192; T <= 288; T += 32) {
      const double c = (double)ceil_div(ceil_div(n_dst, T) * ncol64, slots) * (T * ppr + 8.0 * K);
      if (best_cost == 0 || c < best_cost) { best = T; best_cost = c; }
    }
    return best;
  }
  const int nctt = conv_nctt(cout);
  const int64_t ncol = ceil_div(cout, 16 * nctt);
  auto cost = [&](int T) {
    const int64_t wgs = ceil_div(n_dst, T) * ncol;
    return (double)ceil_div(wgs, slots) * (T * ppr + 8.0 * K);
  };
  // many waves of workgroups and few pairs per row (strides 1/2: 4-5.5 pairs per row, 1.28-1.36x MFMA padding at
  // 128 rows): one 8-wave workgroup per CU on 256..384-row tiles pads 1.12-1.17x (measured +3..8 %)
  if (ceil_div(n_dst, 128) * ncol >= 8 * slots && ppr < 6.5) {
    int T = 384;  // the tallest tile the LDS holds, up to 384 rows (256 -> 384 rows at 96 columns: another +4 %)
    while (T > 128 && (size_t)((T + 1) * (16 * nctt + 4)) * 4 + 1024 > kMaxDynLds) T -= 32;
    if (T >= 192) return T;
  }
  // measured: beyond ~4 waves the choice among 96..160 is within +-3 % either way -> keep the default there
  if (ceil_div(n_dst, 128) * ncol >= 4 * slots) return 128;
  int best = 128;
  double best_cost = cost(128) * 0.95;
  for (int T = 80; T <= 160; T += 16) {
    if (T == 128) continue;
    const size_t lds = (size_t)((T + 1) * (16 * nctt + 4)) * 4 + 5 * 33 * 4 + 16;
    if (2 * (lds + 1024) > 160 * 1024) continue;  // keep two workgroups per CU
    const double c = cost(T);
    if (c < best_cost) { best = T; best_cost = c; }
  }
  return best;
}

// 1 when the fused convolution of this shape can leave per-tile BatchNorm partial sums in its write-back (the wave
// kernels; the tile must hold the reduction scratch). dtype 0: fp32 kernels, 1 / 2: half kernels.
extern "C" int32_t pcs_conv_emits_bn_partials(int32_t cin, int32_t cout, int32_t K, int32_t tile_rows, int32_t dtype) {
  if (cin <= 0 || cout <= 0 || K <= 0 || K > 32 || (cin % 4) || (cout % 4) || tile_rows < 16) return 0;
  int nctt = conv_nctt(cout), nt = 256;
  if (dtype == 0) {
    if (conv5_applies(cin, cout, K)) {
      if (cout >= 192 && tile_rows >= 192) nctt = 4;
      const size_t lds = (size_t)((tile_rows + 1) * (16 * nctt + 4)) * 4 + 1024;
      nt = 2 * lds > 160 * 1024 ? 512 : 256;
    } else if (tile_rows != 64 && tile_rows != 128) {
      return 0;
    }
  } else {
    if (!convh_applies(cin, cout, K)) return 0;
    const size_t lds = (size_t)((tile_rows + 1) * (16 * nctt + 4)) * 4 + 1024;
    nt = 2 * lds > 160 * 1024 ? 512 : 256;
  }
  return conv_stats_fit(tile_rows, 16 * nctt, nt) ? 1 : 0;
}

extern "C" int pcs_conv_gather_gemm_f32(const float *src, int64_t n_src, int32_t cin,
                                        const float *W, int32_t K, int32_t cout,
                                        const int32_t *pairs, int32_t src_col,
                                        const int32_t *seg, int32_t tile_rows, int64_t n_dst,
                                        const float *bias, float *dst, double *bn_partial, void *stream) {
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
  if (bn_partial && !pcs_conv_emits_bn_partials(cin, cout, K, tile_rows, 0)) {
    set_error("pcs_conv_gather_gemm_f32: this shape / tile height does not produce BatchNorm partials (ask pcs_conv_emits_bn_partials)");
    return PCS_EUNSUPPORTED;
  }
  static const int xcd = getenv("PCS_CONV_XCD") ? atoi(getenv("PCS_CONV_XCD")) : 1;  // 0: no XCD-contiguous tile order (debug)
  a.xcd_remap = xcd;
  const bool vec = (cin % 4 == 0) && (cout % 4 == 0) && (((uintptr_t)src | (uintptr_t)W | (uintptr_t)dst | (uintptr_t)bias) & 15) == 0;
  hipStream_t st = as_stream(stream);
  static const int generic = getenv("PCS_CONV_V1") ? atoi(getenv("PCS_CONV_V1")) : 0;  // 1: generic kernel only (debug)
  if (bn_partial && (!vec || generic || (!conv5_applies(cin, cout, K) && K > 32))) {
    set_error("pcs_conv_gather_gemm_f32: BatchNorm partials need the 16-byte-granular wave kernels");
    return PCS_EUNSUPPORTED;
  }
  if (vec && !generic && conv5_applies(cin, cout, K)) return launch_conv_wave5(a, st);
  if (tile_rows != 64 && tile_rows != 128) { set_error("pcs_conv_gather_gemm_f32: this shape takes tile_rows 64 or 128"); return PCS_EUNSUPPORTED; }
  if (vec && !generic && K <= 32) return launch_conv_wave4(a, st);
  return launch_conv_block(a, vec, st);
}
