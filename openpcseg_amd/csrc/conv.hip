// Sparse 3-D convolution on gfx950: output-stationary fused gather-GEMM-accumulate (forward,
// dgrad, transposed) and split-reduction wgrad, fp32 storage + fp32 MFMA (v_mfma_f32_16x16x4_f32).
//
// Reference dataflow being replaced (TS:torchsparse/backend/convolution/convolution_cuda.cu):
// per kernel offset k a gather kernel (:14-24), a cuBLAS mm_out (:149) and a scatter kernel
// (:27-37), i.e. 3 launches and 2 extra HBM round trips of the gathered tile per offset.
//
// Here one workgroup owns `T` consecutive destination rows x a 32*CG column tile and keeps the
// fp32 accumulator tile in LDS for the whole kernel. For each offset k the pairs whose
// destination falls in the tile form ONE contiguous rulebook slice (pairs are sorted by
// destination inside an offset), so the workgroup
//   1. reads the slice (src row, dst row) -> LDS,
//   2. gathers the m src rows (cin chunk of 32) into a compact LDS tile with coalesced 16 B
//      loads (a 32-channel fp32 row chunk = one 128 B line),
//   3. stages the W[k] chunk (32 x CT) in LDS,
//   4. runs 16x16x4 fp32 MFMAs on the compact m x 32 tile (only ceil(m/16) row blocks issue),
//   5. adds the compact result rows into the accumulator tile through the dst-row map.
// Every destination row is written exactly once at the end: no atomics, no zero fill of dst,
// bit-reproducible run to run.
#include "pcs_common.h"

using namespace pcs;

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int CK = 32;        // cin chunk staged per step
constexpr int AS = CK + 2;    // abuf row stride (words): rows*2+g distinct mod 32 -> no conflicts

struct ConvArgs {
  const float *src;
  const float *W;
  const float *bias;
  float *dst;
  const int32_t *pairs;
  const int32_t *seg;
  int64_t n_dst;
  int64_t ntiles;
  int cin, cout, K, src_col, ncoltiles;
};

template <int CG, int RG, int T>
struct ConvCfg {
  static constexpr int CT = 32 * CG;
  static constexpr int NW = CG * RG;
  static constexpr int NT = 64 * NW;
  static constexpr int ACS = CT + 4;   // accumulator row stride
  static constexpr int WS = CT + 16;   // wbuf row stride: == 16 (mod 32)
  static constexpr int NRB = T / 16 / RG;
  static constexpr size_t lds_bytes =
      (size_t)(T * ACS + T * AS + CK * WS) * 4 + (size_t)2 * T * 4;
};

template <int CG, int RG, int T, bool VEC>
__global__ void __launch_bounds__(64 * CG * RG) conv_os_kernel(ConvArgs a) {
  using C = ConvCfg<CG, RG, T>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float *acc_l = reinterpret_cast<float *>(smem);         // [T][ACS]
  float *wbuf = acc_l + T * C::ACS;                       // [CK][WS]   (16 B aligned: T*ACS*4 % 16 == 0)
  float *abuf = wbuf + CK * C::WS;                        // [T][AS]
  int *sidx = reinterpret_cast<int *>(abuf + T * AS);     // [T]
  int *drow = sidx + T;                                   // [T]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = tid >> 6;
  const int cg = wid % CG;
  const int rg = wid / CG;
  const int64_t tile = blockIdx.x / a.ncoltiles;
  const int ctile = blockIdx.x % a.ncoltiles;
  const int n0 = ctile * C::CT;
  const int64_t row0 = tile * T;
  const int64_t nt1 = a.ntiles + 1;

  for (int i = tid; i < T * C::ACS; i += C::NT) acc_l[i] = 0.f;

  for (int k = 0; k < a.K; ++k) {
    const int s = a.seg[(int64_t)k * nt1 + tile];
    const int m = a.seg[(int64_t)k * nt1 + tile + 1] - s;
    if (m <= 0) continue;  // block-uniform
    __syncthreads();       // previous offset fully consumed abuf/wbuf/sidx/drow (and acc zeroing)
    if (tid < m) {
      const int2 p = reinterpret_cast<const int2 *>(a.pairs)[s + tid];
      sidx[tid] = a.src_col ? p.y : p.x;
      drow[tid] = (int)((a.src_col ? p.x : p.y) - row0);
    }
    f32x4 acc[C::NRB][2];
#pragma unroll
    for (int r = 0; r < C::NRB; ++r) { acc[r][0] = (f32x4){0, 0, 0, 0}; acc[r][1] = (f32x4){0, 0, 0, 0}; }

    const float *Wk = a.W + (int64_t)k * a.cin * a.cout;
    for (int c0 = 0; c0 < a.cin; c0 += CK) {
      __syncthreads();  // sidx visible / previous chunk's compute done
      // ---- stage A: m gathered rows x CK channels ------------------------------------------
      if (VEC) {
        for (int e = tid; e < m * (CK / 4); e += C::NT) {
          const int r = e >> 3, c4 = (e & 7) * 4;
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          if (c0 + c4 < a.cin)
            v = *reinterpret_cast<const float4 *>(a.src + (int64_t)sidx[r] * a.cin + c0 + c4);
          float2 *d = reinterpret_cast<float2 *>(abuf + r * AS + c4);
          d[0] = make_float2(v.x, v.y);
          d[1] = make_float2(v.z, v.w);
        }
      } else {
        for (int e = tid; e < m * CK; e += C::NT) {
          const int r = e >> 5, c = e & 31;
          abuf[r * AS + c] = (c0 + c < a.cin) ? a.src[(int64_t)sidx[r] * a.cin + c0 + c] : 0.f;
        }
      }
      // ---- stage W chunk: CK rows x CT cols -------------------------------------------------
      if (VEC) {
        for (int e = tid; e < CK * (C::CT / 4); e += C::NT) {
          const int kr = e / (C::CT / 4), cq = (e % (C::CT / 4)) * 4;
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          if (c0 + kr < a.cin && n0 + cq < a.cout)
            v = *reinterpret_cast<const float4 *>(Wk + (int64_t)(c0 + kr) * a.cout + n0 + cq);
          *reinterpret_cast<float4 *>(wbuf + kr * C::WS + cq) = v;
        }
      } else {
        for (int e = tid; e < CK * C::CT; e += C::NT) {
          const int kr = e / C::CT, cq = e % C::CT;
          wbuf[kr * C::WS + cq] = (c0 + kr < a.cin && n0 + cq < a.cout)
                                      ? Wk[(int64_t)(c0 + kr) * a.cout + n0 + cq] : 0.f;
        }
      }
      __syncthreads();
      // ---- MFMA on the compact tile -----------------------------------------------------------
      const int g = lane >> 4, l15 = lane & 15;
#pragma unroll
      for (int r = 0; r < C::NRB; ++r) {
        const int rb = rg + r * RG;
        if (rb * 16 < m) {  // wave-uniform
          const float *ap = abuf + (rb * 16 + l15) * AS + g;
          const float *bp = wbuf + g * C::WS + cg * 32 + l15;
#pragma unroll
          for (int kk = 0; kk < CK / 4; ++kk) {
            const float av = ap[kk * 4];
            const float b0 = bp[kk * 4 * C::WS];
            const float b1 = bp[kk * 4 * C::WS + 16];
            acc[r][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b0, acc[r][0], 0, 0, 0);
            acc[r][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b1, acc[r][1], 0, 0, 0);
          }
        }
      }
    }
    // ---- add the compact rows into the accumulator tile -------------------------------------
    {
      const int g = lane >> 4, l15 = lane & 15;
#pragma unroll
      for (int r = 0; r < C::NRB; ++r) {
        const int rb = rg + r * RG;
        if (rb * 16 < m) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int cr = rb * 16 + g * 4 + j;
            if (cr < m) {
              float *d = acc_l + drow[cr] * C::ACS + cg * 32 + l15;
              d[0] += acc[r][0][j];
              d[16] += acc[r][1][j];
            }
          }
        }
      }
    }
  }
  __syncthreads();
  // ---- epilogue: write each dst row once --------------------------------------------------------
  const int rows = (int)((a.n_dst - row0) < (int64_t)T ? (a.n_dst - row0) : (int64_t)T);
  if (VEC) {
    for (int e = tid; e < rows * (C::CT / 4); e += C::NT) {
      const int r = e / (C::CT / 4), cq = (e % (C::CT / 4)) * 4;
      if (n0 + cq < a.cout) {
        float4 v = *reinterpret_cast<const float4 *>(acc_l + r * C::ACS + cq);
        if (a.bias) {
          const float4 b = *reinterpret_cast<const float4 *>(a.bias + n0 + cq);
          v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
        }
        *reinterpret_cast<float4 *>(a.dst + (row0 + r) * a.cout + n0 + cq) = v;
      }
    }
  } else {
    for (int e = tid; e < rows * C::CT; e += C::NT) {
      const int r = e / C::CT, cq = e % C::CT;
      if (n0 + cq < a.cout) {
        float v = acc_l[r * C::ACS + cq];
        if (a.bias) v += a.bias[n0 + cq];
        a.dst[(row0 + r) * a.cout + n0 + cq] = v;
      }
    }
  }
}

template <int CG, int RG, int T>
int launch_conv(const ConvArgs &a, bool vec, hipStream_t st) {
  using C = ConvCfg<CG, RG, T>;
  const int64_t nblocks = a.ntiles * a.ncoltiles;
  if (nblocks <= 0) return PCS_OK;
  if (nblocks > 0x7FFFFFFF) { set_error("pcs_conv: grid too large"); return PCS_EUNSUPPORTED; }
  auto kern = vec ? conv_os_kernel<CG, RG, T, true> : conv_os_kernel<CG, RG, T, false>;
  static bool attr_set_v = false, attr_set_s = false;
  bool &flag = vec ? attr_set_v : attr_set_s;
  if (!flag) {
    hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::lds_bytes);
    flag = true;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)nblocks), dim3(C::NT), C::lds_bytes, st, a);
  return check_launch("pcs_conv_gather_gemm_f32");
}

// ================================================================================================
// wgrad:  gW[k] = sum_p fa[ia_p]^T (x) fb[ib_p]
// Work item = (offset k, split s of kPairsPerSplit-ish pairs, 128x128 tile of (ca, cb)).
// Each workgroup gathers 32 pairs at a time into LDS (both operands, whole 128 B lines),
// contracts over the pair axis with 16x16x4 fp32 MFMAs (A = fa^T: lane(i=channel, k=pair)),
// keeps a 128x128 partial in registers (4 waves x 4x4 tiles x 4 regs) and writes it once to the
// workspace; a second kernel sums the splits of each k in a fixed order (deterministic).
// ================================================================================================
constexpr int WG_PB = 32;          // pairs per LDS sub-chunk
constexpr int WG_TS = 128 + 16;    // LDS row stride (== 16 mod 32)

struct WgradArgs {
  const float *fa;
  const float *fb;
  const int32_t *pairs;
  const int32_t *koff;
  float *partial;  // [nsplit_total][ca][cb]
  int ca, cb, K, a_col, pch;
};

__device__ __forceinline__ void find_split(const int32_t *koff, int K, int pch, int split, int *k_out,
                                           int *beg, int *end) {
  int acc = 0;
  for (int k = 0; k < K; ++k) {
    const int nk = koff[k + 1] - koff[k];
    const int ns = (nk + pch - 1) / pch;
    if (split < acc + ns) {
      const int s = split - acc;
      *k_out = k;
      *beg = koff[k] + s * pch;
      const int e = *beg + pch;
      *end = e < koff[k + 1] ? e : koff[k + 1];
      return;
    }
    acc += ns;
  }
  *k_out = -1; *beg = 0; *end = 0;
}

template <bool VEC>
__global__ void __launch_bounds__(256) wgrad_kernel(WgradArgs w) {
  __shared__ __attribute__((aligned(16))) float abuf[WG_PB * WG_TS];
  __shared__ __attribute__((aligned(16))) float bbuf[WG_PB * WG_TS];
  __shared__ int ia[WG_PB], ib[WG_PB];
  __shared__ int sh[3];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int a0 = blockIdx.y * 128, b0 = blockIdx.z * 128;
  const int cat = (w.ca - a0) < 128 ? (w.ca - a0) : 128;  // valid channels in this tile
  const int cbt = (w.cb - b0) < 128 ? (w.cb - b0) : 128;
  const int ta_n = (cat + 15) / 16, tb_n = (cbt + 15) / 16;
  if (tid == 0) find_split(w.koff, w.K, w.pch, blockIdx.x, &sh[0], &sh[1], &sh[2]);
  __syncthreads();
  const int beg = sh[1], end = sh[2];
  const int wa = wid >> 1, wb = wid & 1;  // 2x2 waves; wave owns tiles ta = wa + 2*i, tb = wb + 2*j
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0, 0, 0, 0};

  const int capad = ta_n * 16, cbpad = tb_n * 16;
  for (int p0 = beg; p0 < end; p0 += WG_PB) {
    const int np = (end - p0) < WG_PB ? (end - p0) : WG_PB;
    __syncthreads();
    if (tid < WG_PB) {
      int2 p = make_int2(-1, -1);
      if (tid < np) p = reinterpret_cast<const int2 *>(w.pairs)[p0 + tid];
      ia[tid] = w.a_col ? p.y : p.x;
      ib[tid] = w.a_col ? p.x : p.y;
    }
    __syncthreads();
    if (VEC) {
      for (int e = tid; e < WG_PB * (capad / 4); e += 256) {
        const int r = e / (capad / 4), c4 = (e % (capad / 4)) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < np && c4 < cat) v = *reinterpret_cast<const float4 *>(w.fa + (int64_t)ia[r] * w.ca + a0 + c4);
        *reinterpret_cast<float4 *>(abuf + r * WG_TS + c4) = v;
      }
      for (int e = tid; e < WG_PB * (cbpad / 4); e += 256) {
        const int r = e / (cbpad / 4), c4 = (e % (cbpad / 4)) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < np && c4 < cbt) v = *reinterpret_cast<const float4 *>(w.fb + (int64_t)ib[r] * w.cb + b0 + c4);
        *reinterpret_cast<float4 *>(bbuf + r * WG_TS + c4) = v;
      }
    } else {
      for (int e = tid; e < WG_PB * capad; e += 256) {
        const int r = e / capad, c = e % capad;
        abuf[r * WG_TS + c] = (r < np && c < cat) ? w.fa[(int64_t)ia[r] * w.ca + a0 + c] : 0.f;
      }
      for (int e = tid; e < WG_PB * cbpad; e += 256) {
        const int r = e / cbpad, c = e % cbpad;
        bbuf[r * WG_TS + c] = (r < np && c < cbt) ? w.fb[(int64_t)ib[r] * w.cb + b0 + c] : 0.f;
      }
    }
    __syncthreads();
    const int g = lane >> 4, l15 = lane & 15;
#pragma unroll
    for (int kk = 0; kk < WG_PB / 4; ++kk) {
      float av[4], bv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ta = wa + 2 * i;
        av[i] = (ta < ta_n) ? abuf[(kk * 4 + g) * WG_TS + ta * 16 + l15] : 0.f;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int tb = wb + 2 * j;
        bv[j] = (tb < tb_n) ? bbuf[(kk * 4 + g) * WG_TS + tb * 16 + l15] : 0.f;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (wa + 2 * i < ta_n) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (wb + 2 * j < tb_n)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[j], acc[i][j], 0, 0, 0);
          }
        }
      }
    }
  }
  // write the partial: D[row = channel a][col = channel b]
  float *out = w.partial + (int64_t)blockIdx.x * w.ca * w.cb;
  const int g = lane >> 4, l15 = lane & 15;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int ta = wa + 2 * i;
    if (ta >= ta_n) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int tb = wb + 2 * j;
      if (tb >= tb_n) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ra = ta * 16 + g * 4 + r, cbv = tb * 16 + l15;
        if (ra < cat && cbv < cbt) out[(int64_t)(a0 + ra) * w.cb + b0 + cbv] = acc[i][j][r];
      }
    }
  }
}

// gW[k][e] = sum over the splits of k, in ascending split order
__global__ void __launch_bounds__(256) wgrad_reduce_kernel(const float *__restrict__ partial,
                                                           const int32_t *__restrict__ koff,
                                                           int K, int pch, int64_t cc,
                                                           float *__restrict__ gW) {
  const int k = blockIdx.y;
  __shared__ int sh[2];
  if (threadIdx.x == 0) {
    int acc = 0;
    for (int q = 0; q < k; ++q) acc += (koff[q + 1] - koff[q] + pch - 1) / pch;
    sh[0] = acc;
    sh[1] = (koff[k + 1] - koff[k] + pch - 1) / pch;
  }
  __syncthreads();
  const int base = sh[0], ns = sh[1];
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < cc;
       e += (int64_t)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int q = 0; q < ns; ++q) s += partial[(int64_t)(base + q) * cc + e];
    gW[(int64_t)k * cc + e] = s;
  }
}

int wgrad_plan(const int32_t *koff_host, int K, int *pch_out) {
  // aim for ~2048 splits in total, at least 256 pairs per split
  int64_t P = koff_host[K] - koff_host[0];
  int pch = (int)ceil_div(P > 0 ? P : 1, 2048);
  pch = (int)(ceil_div(pch, WG_PB) * WG_PB);
  if (pch < 256) pch = 256;
  int64_t ns = 0;
  for (int k = 0; k < K; ++k) ns += ceil_div((int64_t)koff_host[k + 1] - koff_host[k], pch);
  *pch_out = pch;
  return (int)ns;
}

}  // namespace

extern "C" int32_t pcs_conv_tile_rows(int32_t cin, int32_t cout) {
  (void)cin;
  (void)cout;
  return 128;
}

extern "C" int pcs_conv_gather_gemm_f32(const float *src, int64_t n_src, int32_t cin,
                                        const float *W, int32_t K, int32_t cout,
                                        const int32_t *pairs, int32_t src_col,
                                        const int32_t *seg, int32_t tile_rows, int64_t n_dst,
                                        const float *bias, float *dst, void *stream) {
  if (cin <= 0 || cout <= 0 || K <= 0 || n_dst < 0 || n_src < 0 || (src_col != 0 && src_col != 1)) {
    set_error("pcs_conv_gather_gemm_f32: bad sizes");
    return PCS_EINVAL;
  }
  if (n_dst == 0) return PCS_OK;
  if (!W || !seg || !dst || (n_src > 0 && !src)) { set_error("pcs_conv_gather_gemm_f32: null pointer"); return PCS_EINVAL; }
  if (tile_rows != 64 && tile_rows != 128) { set_error("pcs_conv_gather_gemm_f32: tile_rows must be 64 or 128"); return PCS_EINVAL; }
  ConvArgs a;
  a.src = src; a.W = W; a.bias = bias; a.dst = dst; a.pairs = pairs; a.seg = seg;
  a.n_dst = n_dst; a.ntiles = ceil_div(n_dst, tile_rows);
  a.cin = cin; a.cout = cout; a.K = K; a.src_col = src_col;
  const bool vec = (cin % 4 == 0) && (cout % 4 == 0) && (((uintptr_t)src | (uintptr_t)W | (uintptr_t)dst | (uintptr_t)bias) & 15) == 0;
  hipStream_t st = as_stream(stream);
  // column tile: 32*CG with CG in 1..4; wider outputs are covered by several column tiles
  int cg = (cout + 31) / 32;
  if (cg > 4) cg = 4;
  a.ncoltiles = (int)ceil_div(cout, 32 * cg);
#define PCS_CONV_CASE(CGv, RGv)                                                       \
  case CGv:                                                                           \
    return tile_rows == 128 ? launch_conv<CGv, RGv, 128>(a, vec, st)                  \
                            : launch_conv<CGv, RGv, 64>(a, vec, st);
  switch (cg) {
    PCS_CONV_CASE(1, 4)
    PCS_CONV_CASE(2, 2)
    PCS_CONV_CASE(3, 2)
    PCS_CONV_CASE(4, 2)
  }
#undef PCS_CONV_CASE
  set_error("pcs_conv_gather_gemm_f32: unreachable");
  return PCS_EINVAL;
}

extern "C" size_t pcs_conv_wgrad_ws_bytes(const int32_t *koff_host, int32_t K, int32_t ca,
                                          int32_t cb) {
  if (!koff_host || K <= 0 || ca <= 0 || cb <= 0) return 0;
  int pch;
  const int ns = wgrad_plan(koff_host, K, &pch);
  return (size_t)(ns > 0 ? ns : 1) * ca * cb * sizeof(float);
}

extern "C" int pcs_conv_wgrad_f32(const float *fa, int32_t ca, const float *fb, int32_t cb,
                                  const int32_t *pairs, int32_t a_col, const int32_t *koff_dev,
                                  const int32_t *koff_host, int32_t K, float *gW, void *ws,
                                  size_t ws_bytes, void *stream) {
  if (ca <= 0 || cb <= 0 || K <= 0 || !koff_dev || !koff_host || !gW || (a_col != 0 && a_col != 1)) {
    set_error("pcs_conv_wgrad_f32: bad args");
    return PCS_EINVAL;
  }
  hipStream_t st = as_stream(stream);
  int pch;
  const int ns = wgrad_plan(koff_host, K, &pch);
  const int64_t cc = (int64_t)ca * cb;
  if (ns == 0) {
    if (hipMemsetAsync(gW, 0, (size_t)K * cc * 4, st) != hipSuccess) { set_error("pcs_conv_wgrad_f32: memset failed"); return PCS_ELAUNCH; }
    return PCS_OK;
  }
  if (!fa || !fb || !pairs || !ws) { set_error("pcs_conv_wgrad_f32: null pointer"); return PCS_EINVAL; }
  if (ws_bytes < (size_t)ns * cc * 4) { set_error("pcs_conv_wgrad_f32: workspace too small"); return PCS_EWORKSPACE; }
  WgradArgs w;
  w.fa = fa; w.fb = fb; w.pairs = pairs; w.koff = koff_dev; w.partial = reinterpret_cast<float *>(ws);
  w.ca = ca; w.cb = cb; w.K = K; w.a_col = a_col; w.pch = pch;
  const bool vec = (ca % 4 == 0) && (cb % 4 == 0) && (((uintptr_t)fa | (uintptr_t)fb) & 15) == 0;
  dim3 grid((unsigned)ns, (unsigned)ceil_div(ca, 128), (unsigned)ceil_div(cb, 128));
  if (vec) hipLaunchKernelGGL(wgrad_kernel<true>, grid, dim3(256), 0, st, w);
  else hipLaunchKernelGGL(wgrad_kernel<false>, grid, dim3(256), 0, st, w);
  int rc = check_launch("pcs_conv_wgrad_f32");
  if (rc) return rc;
  int gx = (int)ceil_div(cc, 256);
  if (gx > 64) gx = 64;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(gx, K), dim3(256), 0, st,
                     reinterpret_cast<const float *>(ws), koff_dev, (int)K, pch, cc, gW);
  return check_launch("pcs_conv_wgrad_f32(reduce)");
}
