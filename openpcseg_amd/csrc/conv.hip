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
#include <stdlib.h>

#include <type_traits>

#include "pcs_common.h"

using namespace pcs;

// debug-build switches (tools/conv_microbench.py through PCS_LIB_PATH); 0 in the product
#ifndef PCS_ABLATE
#define PCS_ABLATE 0   /* v4: 1 no commit, 2 no MFMA, 3 no operand loads in the channel loop */
#endif
#ifndef PCS_TRACE
#define PCS_TRACE 0    /* v5: per-wave phase timers (wall clock, 10 ns) into g_conv_trace; see tools/conv_trace.py */
#endif
#ifndef PCS_ABLATE5
#define PCS_ABLATE5 0  /* v5: 2 no MFMA, 3 no operand loads in the channel loop, 5 no W loads, 6 no A loads */
#endif
#ifndef PCS_ALIAS
#define PCS_ALIAS 0    /* v5: 1 every offset reads W[0], 2 A rows read sequentially instead of gathered */
#endif

namespace {

#if PCS_TRACE
__device__ long long *g_conv_trace;   // [block][wave][8]: t_entry, t_start, t_end, loop, ticket, commit, groups, t_exit
constexpr int kTraceBlocks = 8192;
#define PCS_T(...) __VA_ARGS__
#else
#define PCS_T(...)
#endif

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
  int cin, cout, K, src_col, ncoltiles, xcd_remap, tile_rows;
};

constexpr size_t kMaxDynLds = 160 * 1024 - 256;  // per-workgroup LDS ceiling of a gfx950 CU, minus the static part

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
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::lds_bytes);
    flag = true;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)nblocks), dim3(C::NT), C::lds_bytes, st, a);
  return check_launch("pcs_conv_gather_gemm_f32");
}

// ================================================================================================
// v4: wave-autonomous output-stationary conv (the default for 16-byte-granular shapes).
// PMC on v2/v3 (profiles/round1_conv_pmc.md): ~20 scalar+vector instructions per MFMA, half of all
// wave cycles in s_waitcnt/s_barrier, MFMA pipe 20-28 % busy -- the block-synchronous
// stage->barrier->MFMA step is too small (m ~ 15 rows per offset at 0.05 m LiDAR sparsity).
// Here the only shared state is the fp32 accumulator tile in LDS:
//   * work item = one 16-row block of ONE offset's compact slice; the 8 waves of a workgroup
//     walk the tile's row blocks round-robin with NO barrier in the main loop;
//   * the wave reads its 16 (src,dst) pairs straight from the rulebook (128 B), gathers its A
//     rows from HBM directly in MFMA operand layout (one 16-byte load per lane per 16
//     channels), and reads the W[k] operand straight from L2 with 16-byte loads: lane (g, n)
//     holds W[16j+4g+e][64c+4n .. +3], i.e. B operands of FOUR 16-column tiles whose columns
//     are interleaved (tile f owns columns 4n+f) -- 9 VMEM instructions per 32 MFMAs;
//   * results are added into the LDS tile with ds_add_f32 through the dst-row map, rows of
//     different offsets may interleave in any order (sum order = fp32 rounding noise only);
//   * LDS holds nothing but the accumulator tile -> 2-3 workgroups (16-24 waves) per CU, the
//     gather latency is hidden by wave-level parallelism instead of a software pipeline.
// ================================================================================================
template <int NCTT, int T, int NW_>
struct Conv4Cfg {
  static constexpr int NW = NW_;
  static constexpr int NT = 64 * NW;
  static constexpr int CT = 16 * NCTT;
  static constexpr int ACS = CT + 4;
  static constexpr int N4 = NCTT / 4;            // 64-column groups  (float4 W loads)
  static constexpr int N2 = (NCTT % 4) / 2;      // one 32-column group (float2 W loads)
  static constexpr int N1 = NCTT % 2;            // one 16-column group (float  W loads)
  static constexpr size_t lds_bytes = (size_t)((T + 1) * ACS) * 4 + 4 * 32 * 4 + 32;
};

template <int NCTT, int T, bool E32, int NW, int MINW>
__global__ void __launch_bounds__(64 * NW, MINW) conv_os4_kernel(ConvArgs a) {
  using C = Conv4Cfg<NCTT, T, NW>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float *acc_l = reinterpret_cast<float *>(smem);            // [T+1][ACS], row T = sink for padding rows
  int *kl_k = reinterpret_cast<int *>(acc_l + (T + 1) * C::ACS);  // [32] offset id
  int *kl_s = kl_k + 32;                                     // [32] first pair
  int *kl_m = kl_s + 32;                                     // [32] #pairs
  int *kl_r = kl_m + 32;                                     // [32] first row block (prefix)
  int *commit = kl_r + 33;                                   // ticket: number of row blocks committed
  __shared__ int nk_s;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int g = lane >> 4, l15 = lane & 15;
  // XCD-aware tile mapping: workgroup b runs on XCD b % 8 (observed dispatch order, speed only).
  // Give every XCD one CONTIGUOUS range of tiles so that neighbouring tiles -- which gather
  // overlapping src rows -- share that XCD's L2 (bijective remap for any grid size).
  unsigned bid = blockIdx.x;
  if (a.xcd_remap) {
    const unsigned nb = gridDim.x, q = nb >> 3, r = nb & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int64_t tile = bid / a.ncoltiles;
  const int ctile = bid % a.ncoltiles;
  const int n0 = ctile * C::CT;
  const int64_t row0 = tile * T;
  const int64_t nt1 = a.ntiles + 1;

  if (wid == 0) {  // non-empty offsets of this tile + prefix of their 16-row blocks
    const int k = lane;
    int s0 = 0, m = 0;
    if (k < a.K) {
      s0 = a.seg[(int64_t)k * nt1 + tile];
      m = a.seg[(int64_t)k * nt1 + tile + 1] - s0;
    }
    const unsigned long long mask = __ballot(m > 0);
    const int nrb = (m + 15) >> 4;
    int incl = nrb;
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(incl, o, 64);
      if (lane >= o) incl += t;
    }
    if (m > 0) {
      const int pos = __popcll(mask & ((1ULL << lane) - 1ULL));
      kl_k[pos] = k; kl_s[pos] = s0; kl_m[pos] = m; kl_r[pos] = incl - nrb;
    }
    const int total = __shfl(incl, 63, 64);
    if (lane == 0) { nk_s = __popcll(mask); kl_r[__popcll(mask)] = total; *commit = 0; }
  }
  for (int i = tid; i < (T + 1) * C::ACS; i += C::NT) acc_l[i] = 0.f;
  __syncthreads();
  const int nk = nk_s;
  const int total_rb = nk > 0 ? kl_r[nk] : 0;

  // Every load below is UNCONDITIONAL (addresses clamped into the tensors, values fixed up with
  // selects): a guarded load makes hipcc branch around it and wait vmcnt(0) per load, which
  // serialises the whole gather (measured: 40 us per row block).
  const int cin4 = a.cin - 4;  // last legal float4 start inside a row
  const int wrmax = a.cin - 1;
  // per-lane column offsets of the W loads, clamped inside the row (columns >= cout only feed
  // accumulator columns that the epilogue never writes)
  int col4[C::N4 > 0 ? C::N4 : 1];
#pragma unroll
  for (int q = 0; q < C::N4; ++q) {
    const int c = 64 * q + 4 * l15;
    col4[q] = (n0 + c + 4 <= a.cout) ? c : 0;
  }
  const int c2 = 64 * C::N4 + 2 * l15;
  const int col2 = (n0 + c2 + 2 <= a.cout) ? c2 : 0;
  const int c1 = 64 * C::N4 + 32 * C::N2 + l15;
  const int col1 = (n0 + c1 < a.cout) ? c1 : 0;

  struct Frag {  // operands of one 16-channel block: A (4 channels of this lane's row) + W rows
    float4 a;
    float4 b4[4][C::N4 > 0 ? C::N4 : 1];
    float2 b2[4];
    float b1[4];
  };
  struct Ctx {  // one row block: where its A rows / W slice live, where its results go
    const float *srow0;
    const float *Wk;
    int dloc;
    bool valid;
  };
  auto load_frag = [&](Frag &f, const Ctx &cx, int c0) {
    const int ca = c0 + 4 * g;
    f.a = *reinterpret_cast<const float4 *>(cx.srow0 + (ca <= cin4 ? ca : cin4));
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int wr = ca + e;  // W row of this lane for MFMA e (rows >= cin meet a zero A value)
      const float *wp = cx.Wk + (int64_t)(wr <= wrmax ? wr : wrmax) * a.cout;
#pragma unroll
      for (int q = 0; q < C::N4; ++q) f.b4[e][q] = *reinterpret_cast<const float4 *>(wp + col4[q]);
      if (C::N2) f.b2[e] = *reinterpret_cast<const float2 *>(wp + col2);
      if (C::N1) f.b1[e] = wp[col1];
    }
  };
  // which (offset, row) does row block rb hold for this lane?  i_hint only moves forward.
  auto locate = [&](int rb, int &i_hint, int &pair_idx, bool &valid) {
    while (kl_r[i_hint + 1] <= rb) ++i_hint;
    const int m = kl_m[i_hint];
    const int rk = (rb - kl_r[i_hint]) * 16 + l15;  // row inside the offset's slice
    valid = rk < m;
    // padding rows re-read the slice's last pair (always in bounds) and go to the sink row
    pair_idx = kl_s[i_hint] + (valid ? rk : m - 1);
  };
  auto make_ctx = [&](Ctx &cx, int2 pr, bool valid, int i_k) {
    cx.srow0 = a.src + (int64_t)(a.src_col ? pr.y : pr.x) * a.cin;
    cx.dloc = valid ? (int)((a.src_col ? pr.x : pr.y) - row0) : T;
    cx.valid = valid;
    cx.Wk = a.W + (int64_t)kl_k[i_k] * a.cin * a.cout + n0;
  };

  int i = 0;
  Ctx cur;
  Frag f0, f1;
  if (wid < total_rb) {
    int pidx; bool v;
    locate(wid, i, pidx, v);
    make_ctx(cur, reinterpret_cast<const int2 *>(a.pairs)[pidx], v, i);
    load_frag(f0, cur, 0);
  }
  for (int rb = wid; rb < total_rb; rb += C::NW) {  // wave-uniform loop, no barrier inside
    // the NEXT row block of this wave: its pair is fetched now, its first operand block at the
    // end of this one, so the pair -> A-row dependent chain never stalls the MFMA stream
    const int rbn = rb + C::NW < total_rb ? rb + C::NW : rb;
    int in = i, pidx_n; bool valid_n;
    locate(rbn, in, pidx_n, valid_n);
    const int2 pr_n = reinterpret_cast<const int2 *>(a.pairs)[pidx_n];

    f32x4 acc[NCTT];
#pragma unroll
    for (int t = 0; t < NCTT; ++t) acc[t] = (f32x4){0, 0, 0, 0};
    const bool valid = cur.valid;
    auto mfma_frag = [&](const Frag &f, int c0) {
#if PCS_ABLATE == 2   /* debug build: consume the operands with one VALU op each, no MFMA */
      float t = f.a.x + f.a.y + f.a.z + f.a.w;
      for (int e = 0; e < 4; ++e) {
        for (int q = 0; q < C::N4; ++q) t += f.b4[e][q].x + f.b4[e][q].y + f.b4[e][q].z + f.b4[e][q].w;
        if (C::N2) t += f.b2[e].x + f.b2[e].y;
        if (C::N1) t += f.b1[e];
      }
      acc[0][0] += t;
      return;
#endif
      const bool aok = valid && (c0 + 4 * g) <= cin4;
      const float ae[4] = {aok ? f.a.x : 0.f, aok ? f.a.y : 0.f, aok ? f.a.z : 0.f, aok ? f.a.w : 0.f};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
#pragma unroll
        for (int q = 0; q < C::N4; ++q) {
          acc[4 * q + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ae[e], f.b4[e][q].x, acc[4 * q + 0], 0, 0, 0);
          acc[4 * q + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ae[e], f.b4[e][q].y, acc[4 * q + 1], 0, 0, 0);
          acc[4 * q + 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(ae[e], f.b4[e][q].z, acc[4 * q + 2], 0, 0, 0);
          acc[4 * q + 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(ae[e], f.b4[e][q].w, acc[4 * q + 3], 0, 0, 0);
        }
        if (C::N2) {
          acc[4 * C::N4 + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ae[e], f.b2[e].x, acc[4 * C::N4 + 0], 0, 0, 0);
          acc[4 * C::N4 + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ae[e], f.b2[e].y, acc[4 * C::N4 + 1], 0, 0, 0);
        }
        if (C::N1) acc[NCTT - 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ae[e], f.b1[e], acc[NCTT - 1], 0, 0, 0);
      }
    };
    Ctx nxt;
    if (E32) {
      // cin % 32 == 0: straight-line body. Two register sets, explicitly software-pipelined;
      // sched_barrier pins "issue the next block's 9 loads, THEN this block's MFMAs" (left alone
      // the machine scheduler sinks each load next to its use and only 1-2 stay in flight);
      // no branch between a load and its use, so every wait is a counted vmcnt.
      // PIPE(load next block, MFMAs of this block): both live in ONE scheduling region and a
      // sched_group_barrier sequence interleaves them -- per contraction step e: the W loads of
      // step e of the NEXT block, then the NCTT MFMAs of step e of THIS block -- so the VMEM issue
      // and its address arithmetic sit in the shadow of the 32-cycle MFMAs instead of in a gap.
#if PCS_ABLATE == 3   /* debug build: no operand loads inside the channel loop */
#define PCS_PIPE(LOAD, MFMA) MFMA; __builtin_amdgcn_sched_barrier(0);
#else
#define PCS_PIPE(LOAD, MFMA)                                                                       \
  LOAD; MFMA;                                                                                      \
  __builtin_amdgcn_sched_group_barrier(0x020, 1 + C::N4 + C::N2 + C::N1, 0);                       \
  __builtin_amdgcn_sched_group_barrier(0x008, NCTT, 0);                                            \
  __builtin_amdgcn_sched_group_barrier(0x020, C::N4 + C::N2 + C::N1, 0);                           \
  __builtin_amdgcn_sched_group_barrier(0x008, NCTT, 0);                                            \
  __builtin_amdgcn_sched_group_barrier(0x020, C::N4 + C::N2 + C::N1, 0);                           \
  __builtin_amdgcn_sched_group_barrier(0x008, NCTT, 0);                                            \
  __builtin_amdgcn_sched_group_barrier(0x020, C::N4 + C::N2 + C::N1, 0);                           \
  __builtin_amdgcn_sched_group_barrier(0x008, NCTT, 0);                                            \
  __builtin_amdgcn_sched_barrier(0);
#endif
      for (int c0 = 0; c0 < a.cin - 32; c0 += 32) {
        PCS_PIPE(load_frag(f1, cur, c0 + 16), mfma_frag(f0, c0))
        PCS_PIPE(load_frag(f0, cur, c0 + 32), mfma_frag(f1, c0 + 16))
      }
      PCS_PIPE(load_frag(f1, cur, a.cin - 16), mfma_frag(f0, a.cin - 32))
      make_ctx(nxt, pr_n, valid_n, in);
      __builtin_amdgcn_sched_barrier(0);
      // first block of the next row block: in flight during the last MFMAs and the commit
      PCS_PIPE(load_frag(f0, nxt, 0), mfma_frag(f1, a.cin - 16))
#undef PCS_PIPE
    } else {
      for (int c0 = 0; c0 < a.cin; c0 += 32) {  // branches are wave-uniform (kernel args)
        const bool has1 = c0 + 16 < a.cin;
        if (has1) load_frag(f1, cur, c0 + 16);
        mfma_frag(f0, c0);
        if (has1) {
          if (c0 + 32 < a.cin) load_frag(f0, cur, c0 + 32);
          mfma_frag(f1, c0 + 16);
        }
      }
      make_ctx(nxt, pr_n, valid_n, in);
      load_frag(f0, nxt, 0);
    }
    const int dloc = cur.dloc;
    cur = nxt;
    i = in;
    // ---- in-order commit -------------------------------------------------------------------------
    // Row blocks of different offsets may hit the same dst row, so the LDS tile update must be
    // exclusive. ds_add_f32 is ~200 cycles per wave-instruction on gfx950 (measured: LDS pipe
    // 93 % busy), so instead each row block commits in ticket order: wait until every earlier
    // row block of the tile has committed, plain ds_read/add/ds_write, publish. Row blocks are
    // numbered offset-major, hence every dst element is summed in ascending-offset order -- the
    // reference's order -- and the result is bit-reproducible.
#if PCS_ABLATE == 1   /* debug build: keep one cheap use of the accumulators, skip the commit */
    { float t = 0.f; for (int q = 0; q < NCTT; ++q) t += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
      if (t == 1.2345e30f) acc_l[lane] = t + (float)dloc; }
    continue;
#endif
    if (lane == 0) {
      while (__hip_atomic_load(commit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != rb)
        __builtin_amdgcn_s_sleep(1);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    // D[row = 4g+j][col = l15] of tile t  ->  accumulator row dloc(4g+j), interleaved column map
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int dr = __shfl(dloc, 4 * g + j, 64);  // dloc of compact row 4g+j lives in lanes l15 == 4g+j
      float *d = acc_l + dr * C::ACS;
#pragma unroll
      for (int q = 0; q < C::N4; ++q) {
        float4 *p4 = reinterpret_cast<float4 *>(d + 64 * q + 4 * l15);
        float4 v = *p4;
        v.x += acc[4 * q + 0][j]; v.y += acc[4 * q + 1][j]; v.z += acc[4 * q + 2][j]; v.w += acc[4 * q + 3][j];
        *p4 = v;
      }
      if (C::N2) {
        float2 *p2 = reinterpret_cast<float2 *>(d + 64 * C::N4 + 2 * l15);
        float2 v = *p2;
        v.x += acc[4 * C::N4 + 0][j]; v.y += acc[4 * C::N4 + 1][j];
        *p2 = v;
      }
      if (C::N1) d[64 * C::N4 + 32 * C::N2 + l15] += acc[NCTT - 1][j];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    if (lane == 0) __hip_atomic_store(commit, rb + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  __syncthreads();
  // ---- epilogue: every dst row written once ---------------------------------------------------------
  const int rows = (int)((a.n_dst - row0) < (int64_t)T ? (a.n_dst - row0) : (int64_t)T);
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
}

template <int NCTT, int T, int NW, int MINW>
int launch_conv4_cfg(const ConvArgs &a, hipStream_t st) {
  using C = Conv4Cfg<NCTT, T, NW>;
  const int64_t nblocks = a.ntiles * a.ncoltiles;
  if (nblocks <= 0) return PCS_OK;
  if (nblocks > 0x7FFFFFFF) { set_error("pcs_conv: grid too large"); return PCS_EUNSUPPORTED; }
  const bool e32 = (a.cin % 32) == 0;
  auto kern = e32 ? conv_os4_kernel<NCTT, T, true, NW, MINW> : conv_os4_kernel<NCTT, T, false, NW, MINW>;
  static bool attr_set[2] = {false, false};
  if (!attr_set[e32]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::lds_bytes);
    attr_set[e32] = true;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)nblocks), dim3(C::NT), C::lds_bytes, st, a);
  return check_launch("pcs_conv_gather_gemm_f32(v4)");
}

template <int NCTT, int T>
int launch_conv4(const ConvArgs &a, hipStream_t st) {
  // workgroup shape: 4 waves with <= 168 VGPRs (default: 3 workgroups / CU) or 8 waves
  static const int nw = getenv("PCS_CONV_NW") ? atoi(getenv("PCS_CONV_NW")) : 4;
  if (nw == 4) return launch_conv4_cfg<NCTT, T, 4, 3>(a, st);
  if (nw == 84) return launch_conv4_cfg<NCTT, T, 8, 4>(a, st);  // 8 waves, <= 128 VGPRs: 2 workgroups = 16 waves / CU
  return launch_conv4_cfg<NCTT, T, 8, 2>(a, st);
}

// ================================================================================================
// v5 = v4 + row-block GROUPS. Ablation of v4 (profiles/round1_conv_pmc.md): with the MFMAs removed the
// kernel still needs 60-77 % of its time -- every 16-row block streams its own copy of W[k] (Cin x CT fp32,
// 37-131 KB) out of L2: 16-19 GB per layer, 12-17 TB/s. 8 flop per W byte cannot be fed by the L2.
// Here one wave applies each W operand block to a GROUP of up to R consecutive row blocks of the same
// offset (R accumulator sets), so W traffic per compact row drops R-fold where an offset has >= R row
// blocks in the tile; the workgroup shape / tile height are chosen per layer so that it usually does.
// Everything else is v4: register-direct operands, interleaved column tiles, sched_group_barrier
// software pipeline, cross-group prefetch, ticket-ordered commit. Requires cin % 32 == 0.
// ================================================================================================
template <int NCTT, int NW_, int R_>
struct Conv5Cfg {
  static constexpr int NW = NW_;
  static constexpr int R = R_;
  static constexpr int NT = 64 * NW;
  static constexpr int CT = 16 * NCTT;
  static constexpr int ACS = CT + 4;
  static constexpr int N4 = NCTT / 4;
  static constexpr int N2 = (NCTT % 4) / 2;
  static constexpr int N1 = NCTT % 2;
  static constexpr int NWL = N4 + N2 + N1;  // W loads per contraction step
  static constexpr size_t lds_bytes(int T) { return (size_t)((T + 1) * ACS) * 4 + 5 * 33 * 4 + 16; }
};

template <int NCTT, int NW, int MINW, int R>
__global__ void __launch_bounds__(64 * NW, MINW) conv_os5_kernel(ConvArgs a) {
  using C = Conv5Cfg<NCTT, NW, R>;
  const int T = a.tile_rows;  // any multiple of 16: the host picks it per layer (pcs_conv_pick_tile_rows)
  PCS_T(const long long tr_entry = wall_clock64(); long long tr_loop = 0, tr_ticket = 0, tr_commit = 0; int tr_groups = 0;)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float *acc_l = reinterpret_cast<float *>(smem);            // [T+1][ACS], row T = sink for padding rows
  int *kl_k = reinterpret_cast<int *>(acc_l + (T + 1) * C::ACS);  // [32] offset id
  int *kl_s = kl_k + 32;                                     // [32] first pair
  int *kl_m = kl_s + 32;                                     // [32] #pairs
  int *kl_g = kl_m + 32;                                     // [33] first FULL group (prefix over the offsets)
  int *kl_h = kl_g + 33;                                     // [33] first partial group (prefix)
  int *commit = kl_h + 33;
  __shared__ int nk_s;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int g = lane >> 4, l15 = lane & 15;
  unsigned bid = blockIdx.x;
  if (a.xcd_remap) {
    const unsigned nb = gridDim.x, q = nb >> 3, r = nb & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int64_t tile = bid / a.ncoltiles;
  const int ctile = bid % a.ncoltiles;
  const int n0 = ctile * C::CT;
  const int64_t row0 = tile * T;
  const int64_t nt1 = a.ntiles + 1;

  if (wid == 0) {  // non-empty offsets of this tile + prefix of their row-block groups
    const int k = lane;
    int s0 = 0, m = 0;
    if (k < a.K) {
      s0 = a.seg[(int64_t)k * nt1 + tile];
      m = a.seg[(int64_t)k * nt1 + tile + 1] - s0;
    }
    const unsigned long long mask = __ballot(m > 0);
    const int nrb = (m + 15) >> 4;
    const int nfull = nrb / R, npart = (nrb % R) ? 1 : 0;  // groups of R row blocks + at most one shorter group
    int incl = nfull | (npart << 16);                       // both prefixes in one scan
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(incl, o, 64);
      if (lane >= o) incl += t;
    }
    if (m > 0) {
      const int pos = __popcll(mask & ((1ULL << lane) - 1ULL));
      kl_k[pos] = k; kl_s[pos] = s0; kl_m[pos] = m;
      kl_g[pos] = (incl & 0xFFFF) - nfull; kl_h[pos] = (incl >> 16) - npart;
    }
    const int total = __shfl(incl, 63, 64);
    if (lane == 0) {
      const int nkk = __popcll(mask);
      nk_s = nkk; kl_g[nkk] = total & 0xFFFF; kl_h[nkk] = total >> 16; *commit = 0;
    }
  }
  for (int i = tid; i < (T + 1) * C::ACS; i += C::NT) acc_l[i] = 0.f;
  __syncthreads();
  const int nk = nk_s;
  // group order = commit order: all full groups (R row blocks, equal duration) in ascending offset order, then the
  // partial groups. Waves take groups round-robin and commit in order, so neighbours of equal length never wait
  // for each other (with offset-major numbering a short group queued behind a long one idled its wave: 9-12 % of
  // the wave time in the ticket wait, tools/conv_trace.py). The order depends on the map only: deterministic.
  const int total_full = nk > 0 ? kl_g[nk] : 0;
  const int total_grp = nk > 0 ? total_full + kl_h[nk] : 0;

  const int cin4 = a.cin - 4;
  int col4[C::N4 > 0 ? C::N4 : 1];
#pragma unroll
  for (int q = 0; q < C::N4; ++q) {
    const int c = 64 * q + 4 * l15;
    col4[q] = (n0 + c + 4 <= a.cout) ? c : 0;
  }
  const int c2 = 64 * C::N4 + 2 * l15;
  const int col2 = (n0 + c2 + 2 <= a.cout) ? c2 : 0;
  const int c1 = 64 * C::N4 + 32 * C::N2 + l15;
  const int col1 = (n0 + c1 < a.cout) ? c1 : 0;

  struct Frag {  // one 16-channel block: A pieces of the R row blocks + the shared W rows
    float4 a[R];
    float4 b4[4][C::N4 > 0 ? C::N4 : 1];
    float2 b2[4];
    float b1[4];
  };
  struct Ctx {  // one group: R row blocks of one offset
    const float *srow0[R];
    const float *Wk;
    int dloc[R];
    int nr;  // row blocks really present (1..R)
    unsigned vmask;  // bit r: this lane's row of block r is a real pair
  };
  // PCS_ABLATE5 (debug builds): 3 = no operand loads inside the channel loop, 5 = no W loads, 6 = no A loads there
  auto load_frag = [&](Frag &f, const Ctx &cx, int c0, bool in_loop = false) {
    const int ca = c0 + 4 * g;  // cin % 32 == 0: always inside the row
    (void)in_loop;
#if PCS_ABLATE5 == 3
    if (in_loop) return;
#endif
#if PCS_ABLATE5 != 6
#pragma unroll
    for (int r = 0; r < R; ++r) f.a[r] = *reinterpret_cast<const float4 *>(cx.srow0[r] + ca);
#else
    if (!in_loop) for (int r = 0; r < R; ++r) f.a[r] = *reinterpret_cast<const float4 *>(cx.srow0[r] + ca);
#endif
#if PCS_ABLATE5 == 5
    if (in_loop) return;
#endif
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float *wp = cx.Wk + (int64_t)(ca + e) * a.cout;
#pragma unroll
      for (int q = 0; q < C::N4; ++q) f.b4[e][q] = *reinterpret_cast<const float4 *>(wp + col4[q]);
      if (C::N2) f.b2[e] = *reinterpret_cast<const float2 *>(wp + col2);
      if (C::N1) f.b1[e] = wp[col1];
    }
  };
  // group grp -> its offset entry (the hint only moves forward inside a phase; bit 5 = partial-group phase),
  // pair index of this lane per row block
  auto locate = [&](int grp, int &i_hint, int *pidx, unsigned &vmask, int &nr) {
    int rb0, e;
    if (grp < total_full) {
      e = i_hint;
      while (kl_g[e + 1] <= grp) ++e;
      i_hint = e;
      rb0 = (grp - kl_g[e]) * R;
      nr = R;
    } else {
      const int q = grp - total_full;
      e = (i_hint & 32) ? (i_hint & 31) : 0;
      while (kl_h[e + 1] <= q) ++e;
      i_hint = e | 32;
      const int nrb = (kl_m[e] + 15) >> 4;
      rb0 = (nrb / R) * R;
      nr = nrb - rb0;
    }
    const int m = kl_m[e];
    vmask = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int rk = (rb0 + r) * 16 + l15;
      const bool v = rk < m;
      vmask |= v ? (1u << r) : 0u;
      pidx[r] = kl_s[e] + (v ? rk : m - 1);  // padding rows re-read the slice's last pair
    }
  };
  auto make_ctx = [&](Ctx &cx, const int2 *pr, unsigned vmask, int nr, int i_k) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
#if PCS_ALIAS == 2   /* debug build: gather replaced by the dst row itself (sequential, L2-friendly A operand) */
      cx.srow0[r] = a.src + (int64_t)(a.src_col ? pr[r].x : pr[r].y) * a.cin;
#else
      cx.srow0[r] = a.src + (int64_t)(a.src_col ? pr[r].y : pr[r].x) * a.cin;
#endif
      cx.dloc[r] = ((vmask >> r) & 1u) ? (int)((a.src_col ? pr[r].x : pr[r].y) - row0) : T;
    }
    cx.vmask = vmask;
    cx.nr = nr;
#if PCS_ALIAS == 1   /* debug build: every offset reads W[0] (W operand always L1/L2-hot) */
    cx.Wk = a.W + n0;
#else
    cx.Wk = a.W + (int64_t)kl_k[i_k & 31] * a.cin * a.cout + n0;
#endif
  };

  int i = 0;
  Ctx cur;
  Frag f0, f1;
  if (wid < total_grp) {
    int pidx[R]; unsigned vm; int nr;
    locate(wid, i, pidx, vm, nr);
    int2 pr[R];
#pragma unroll
    for (int r = 0; r < R; ++r) pr[r] = reinterpret_cast<const int2 *>(a.pairs)[pidx[r]];
    make_ctx(cur, pr, vm, nr, i);
    load_frag(f0, cur, 0);
  }
  PCS_T(const long long tr_start = wall_clock64();)
  for (int grp = wid; grp < total_grp; grp += C::NW) {  // wave-uniform loop, no barrier inside
    PCS_T(const long long tr_a = wall_clock64();)
    const int grpn = grp + C::NW < total_grp ? grp + C::NW : grp;
    int in = i, pidx_n[R], nr_n; unsigned vm_n;
    locate(grpn, in, pidx_n, vm_n, nr_n);
    int2 pr_n[R];
#pragma unroll
    for (int r = 0; r < R; ++r) pr_n[r] = reinterpret_cast<const int2 *>(a.pairs)[pidx_n[r]];

    f32x4 acc[R][NCTT];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int t = 0; t < NCTT; ++t) acc[r][t] = (f32x4){0, 0, 0, 0};
    const unsigned vmask = cur.vmask;
    const int nr = cur.nr;  // wave-uniform
    // MFMAs of one 16-channel block for the first NR row blocks of the group (NR is wave-uniform)
    auto mfma_frag = [&](const Frag &f, auto nr_tag) {
      constexpr int NR = decltype(nr_tag)::value;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          const bool ok = (vmask >> r) & 1u;
          const float av = ok ? (e == 0 ? f.a[r].x : (e == 1 ? f.a[r].y : (e == 2 ? f.a[r].z : f.a[r].w))) : 0.f;
#if PCS_ABLATE5 == 2   /* debug build: consume the operands with one VALU op each, no MFMA */
#pragma unroll
          for (int q = 0; q < C::N4; ++q)
            acc[r][4 * q][0] += av * (f.b4[e][q].x + f.b4[e][q].y + f.b4[e][q].z + f.b4[e][q].w);
          if (C::N2) acc[r][4 * C::N4][0] += av * (f.b2[e].x + f.b2[e].y);
          if (C::N1) acc[r][NCTT - 1][0] += av * f.b1[e];
          continue;
#endif
#pragma unroll
          for (int q = 0; q < C::N4; ++q) {
            acc[r][4 * q + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, f.b4[e][q].x, acc[r][4 * q + 0], 0, 0, 0);
            acc[r][4 * q + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, f.b4[e][q].y, acc[r][4 * q + 1], 0, 0, 0);
            acc[r][4 * q + 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, f.b4[e][q].z, acc[r][4 * q + 2], 0, 0, 0);
            acc[r][4 * q + 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, f.b4[e][q].w, acc[r][4 * q + 3], 0, 0, 0);
          }
          if (C::N2) {
            acc[r][4 * C::N4 + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, f.b2[e].x, acc[r][4 * C::N4 + 0], 0, 0, 0);
            acc[r][4 * C::N4 + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, f.b2[e].y, acc[r][4 * C::N4 + 1], 0, 0, 0);
          }
          if (C::N1) acc[r][NCTT - 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, f.b1[e], acc[r][NCTT - 1], 0, 0, 0);
        }
      }
    };
    Ctx nxt;
    // one scheduling region per block: per contraction step e the W loads of the NEXT block (plus,
    // first, its R A pieces), then this block's NR*NCTT MFMAs of step e
#define PCS_PIPE5(LOAD, FR, NRV)                                                                   \
  LOAD; mfma_frag(FR, std::integral_constant<int, NRV>{});                                         \
  __builtin_amdgcn_sched_group_barrier(0x020, R + C::NWL, 0);                                      \
  __builtin_amdgcn_sched_group_barrier(0x008, NRV * NCTT, 0);                                      \
  __builtin_amdgcn_sched_group_barrier(0x020, C::NWL, 0);                                          \
  __builtin_amdgcn_sched_group_barrier(0x008, NRV * NCTT, 0);                                      \
  __builtin_amdgcn_sched_group_barrier(0x020, C::NWL, 0);                                          \
  __builtin_amdgcn_sched_group_barrier(0x008, NRV * NCTT, 0);                                      \
  __builtin_amdgcn_sched_group_barrier(0x020, C::NWL, 0);                                          \
  __builtin_amdgcn_sched_group_barrier(0x008, NRV * NCTT, 0);                                      \
  __builtin_amdgcn_sched_barrier(0);
#define PCS_BODY5(NRV)                                                                             \
  {                                                                                                \
    for (int c0 = 0; c0 < a.cin - 32; c0 += 32) {                                                  \
      PCS_PIPE5(load_frag(f1, cur, c0 + 16, true), f0, NRV)                                        \
      PCS_PIPE5(load_frag(f0, cur, c0 + 32, true), f1, NRV)                                        \
    }                                                                                              \
    PCS_PIPE5(load_frag(f1, cur, a.cin - 16, true), f0, NRV)                                       \
    make_ctx(nxt, pr_n, vm_n, nr_n, in);                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                             \
    PCS_PIPE5(load_frag(f0, nxt, 0), f1, NRV)                                                      \
  }
    if (R >= 4 && nr == 4) PCS_BODY5((R >= 4 ? 4 : 1))
    else if (R >= 3 && nr == 3) PCS_BODY5((R >= 3 ? 3 : 1))
    else if (R >= 2 && nr == 2) PCS_BODY5((R >= 2 ? 2 : 1))
    else PCS_BODY5(1)
#undef PCS_BODY5
#undef PCS_PIPE5
    // ---- in-order commit of the group's row blocks (see v4) --------------------------------------------
    PCS_T(const long long tr_b = wall_clock64();)
    if (lane == 0) {
      while (__hip_atomic_load(commit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != grp)
        __builtin_amdgcn_s_sleep(1);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    PCS_T(const long long tr_c = wall_clock64();)
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (r < nr) {  // wave-uniform
        // all LDS reads of the block first (one latency), then the adds, then the writes
        float *d[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) d[j] = acc_l + __shfl(cur.dloc[r], 4 * g + j, 64) * C::ACS;
        float4 v4[4][C::N4 > 0 ? C::N4 : 1];
        float2 v2[4];
        float v1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
          for (int q = 0; q < C::N4; ++q) v4[j][q] = *reinterpret_cast<const float4 *>(d[j] + 64 * q + 4 * l15);
          if (C::N2) v2[j] = *reinterpret_cast<const float2 *>(d[j] + 64 * C::N4 + 2 * l15);
          if (C::N1) v1[j] = d[j][64 * C::N4 + 32 * C::N2 + l15];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
          for (int q = 0; q < C::N4; ++q) {
            v4[j][q].x += acc[r][4 * q + 0][j]; v4[j][q].y += acc[r][4 * q + 1][j];
            v4[j][q].z += acc[r][4 * q + 2][j]; v4[j][q].w += acc[r][4 * q + 3][j];
            *reinterpret_cast<float4 *>(d[j] + 64 * q + 4 * l15) = v4[j][q];
          }
          if (C::N2) {
            v2[j].x += acc[r][4 * C::N4 + 0][j]; v2[j].y += acc[r][4 * C::N4 + 1][j];
            *reinterpret_cast<float2 *>(d[j] + 64 * C::N4 + 2 * l15) = v2[j];
          }
          if (C::N1) d[j][64 * C::N4 + 32 * C::N2 + l15] = v1[j] + acc[r][NCTT - 1][j];
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    if (lane == 0) __hip_atomic_store(commit, grp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    PCS_T(const long long tr_d = wall_clock64(); tr_loop += tr_b - tr_a; tr_ticket += tr_c - tr_b; tr_commit += tr_d - tr_c; ++tr_groups;)
    cur = nxt;
    i = in;
  }
  PCS_T(const long long tr_end = wall_clock64();)
  __syncthreads();
  const int rows = (int)((a.n_dst - row0) < (int64_t)T ? (a.n_dst - row0) : (int64_t)T);
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
#if PCS_TRACE
  if (lane == 0 && blockIdx.x < kTraceBlocks && g_conv_trace) {
    long long *t = g_conv_trace + ((int64_t)blockIdx.x * 8 + wid) * 8;
    t[0] = tr_entry; t[1] = tr_start; t[2] = tr_end; t[3] = tr_loop; t[4] = tr_ticket; t[5] = tr_commit;
    t[6] = tr_groups; t[7] = wall_clock64();
  }
#endif
}

#if PCS_TRACE
long long *g_trace_host_ptr = nullptr;
void trace_prepare(hipStream_t st) {
  if (!g_trace_host_ptr) {
    (void)hipMalloc(&g_trace_host_ptr, (size_t)kTraceBlocks * 64 * sizeof(long long));
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_conv_trace), &g_trace_host_ptr, sizeof(g_trace_host_ptr));
  }
  (void)hipMemsetAsync(g_trace_host_ptr, 0, (size_t)kTraceBlocks * 64 * sizeof(long long), st);
}
#endif

template <int NCTT, int NW, int MINW, int R>
int launch_conv5(const ConvArgs &a, hipStream_t st) {
  using C = Conv5Cfg<NCTT, NW, R>;
  const int64_t nblocks = a.ntiles * a.ncoltiles;
  if (nblocks <= 0) return PCS_OK;
  if (nblocks > 0x7FFFFFFF) { set_error("pcs_conv: grid too large"); return PCS_EUNSUPPORTED; }
  auto kern = conv_os5_kernel<NCTT, NW, MINW, R>;
  const size_t lds = C::lds_bytes(a.tile_rows);
  if (lds > kMaxDynLds) { set_error("pcs_conv: tile_rows too large for this column tile"); return PCS_EUNSUPPORTED; }
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxDynLds);
    attr_set = true;
  }
  PCS_T(trace_prepare(st);)
  hipLaunchKernelGGL(kern, dim3((unsigned)nblocks), dim3(C::NT), lds, st, a);
  return check_launch("pcs_conv_gather_gemm_f32(v5)");
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

// Same sum for 16-byte-granular weight blocks, spread over the chip also when K is 1 or 8 (pointwise and
// 2x2x2 layers have hundreds of splits per offset): a workgroup owns 64 consecutive elements of one offset;
// its 16 split lanes each sum every 16th split (4 independent loads in flight), and the 16 partial sums are
// combined in lane order through LDS, so the result does not depend on the launch.
__global__ void __launch_bounds__(256) wgrad_reduce4_kernel(const float *__restrict__ partial,
                                                            const int32_t *__restrict__ koff,
                                                            int K, int pch, int64_t cc,
                                                            float *__restrict__ gW) {
  const int k = blockIdx.y;
  __shared__ int sh[2];
  __shared__ float4 red[16][16];
  if (threadIdx.x == 0) {
    int acc = 0;
    for (int q = 0; q < k; ++q) acc += (koff[q + 1] - koff[q] + pch - 1) / pch;
    sh[0] = acc;
    sh[1] = (koff[k + 1] - koff[k] + pch - 1) / pch;
  }
  __syncthreads();
  const int base = sh[0], ns = sh[1];
  const int et = threadIdx.x & 15, ql = threadIdx.x >> 4;
  const int64_t e = ((int64_t)blockIdx.x * 16 + et) * 4;
  const bool ok = e < cc;
  const float *p = partial + (int64_t)base * cc + (ok ? e : 0);
  float4 s0 = {0, 0, 0, 0}, s1 = s0, s2 = s0, s3 = s0;
  int q = ql;
  for (; q + 48 < ns; q += 64) {
    const float4 v0 = *reinterpret_cast<const float4 *>(p + (int64_t)q * cc);
    const float4 v1 = *reinterpret_cast<const float4 *>(p + (int64_t)(q + 16) * cc);
    const float4 v2 = *reinterpret_cast<const float4 *>(p + (int64_t)(q + 32) * cc);
    const float4 v3 = *reinterpret_cast<const float4 *>(p + (int64_t)(q + 48) * cc);
    s0.x += v0.x; s0.y += v0.y; s0.z += v0.z; s0.w += v0.w;
    s1.x += v1.x; s1.y += v1.y; s1.z += v1.z; s1.w += v1.w;
    s2.x += v2.x; s2.y += v2.y; s2.z += v2.z; s2.w += v2.w;
    s3.x += v3.x; s3.y += v3.y; s3.z += v3.z; s3.w += v3.w;
  }
  for (; q < ns; q += 16) {
    const float4 v0 = *reinterpret_cast<const float4 *>(p + (int64_t)q * cc);
    s0.x += v0.x; s0.y += v0.y; s0.z += v0.z; s0.w += v0.w;
  }
  s0.x += s1.x + (s2.x + s3.x); s0.y += s1.y + (s2.y + s3.y);
  s0.z += s1.z + (s2.z + s3.z); s0.w += s1.w + (s2.w + s3.w);
  red[ql][et] = s0;
  __syncthreads();
  if (ql == 0 && ok) {
    float4 t = red[0][et];
#pragma unroll
    for (int j = 1; j < 16; ++j) {
      const float4 v = red[j][et];
      t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
    }
    *reinterpret_cast<float4 *>(gW + (int64_t)k * cc + e) = t;
  }
}

// ================================================================================================
// wgrad v2: wave-autonomous, operands straight from HBM/L2 into MFMA layout, no LDS, no barrier.
//   gW[k][a][b] = sum over the pairs p of offset k:  fa[ia_p][a] * fb[ib_p][b]
// MFMA 16x16x4 with the PAIR axis as the contraction: lane (n = lane&15, g = lane>>4) holds, for
// pair 4j+g, the 16-byte pieces fa[ia][a0+4n .. +3] and fb[ib][b0+4n .. +3]; component f of the
// A piece and component h of the B piece feed output tile (f,h), whose rows/cols are the
// interleaved channels {a0+4i+f} x {b0+4n+h}. One 64x64 output block = 16 tiles = 16 MFMAs per
// TWO 16-byte loads per lane. A workgroup = 4 waves = 4 output blocks of one (offset, pair chunk)
// split; partial blocks go to the workspace and wgrad_reduce_kernel sums the splits in order.
// ================================================================================================
struct Wgrad2Args {
  const float *fa;
  const float *fb;
  const int32_t *pairs;
  const int32_t *koff;
  float *partial;  // [nsplit_total][ca][cb]
  int ca, cb, K, a_col, pch, nbg;  // nbg = number of b-groups
};

__host__ __device__ inline int wg_ngroups(int c) { return (c + 63) / 64; }
// width of the channel groups of a c-channel operand: c is cut into ceil(c/64) EQUAL groups of 16, 32, 48 or 64
// channels (96 -> 48 + 48, not 64 + 32: the four waves of a workgroup own one output block each, and unequal blocks
// leave three SIMDs waiting for the 64x64 one -- 96-channel layers ran at 55 % of the 256-channel rate)
__host__ __device__ inline int wg_gwidth(int c) {
  const int per = (c + wg_ngroups(c) - 1) / wg_ngroups(c);
  return (per + 15) / 16 * 16;
}

template <int W> struct WVec;
template <> struct WVec<64> { using T = float4; static constexpr int N = 4; };
template <> struct WVec<48> { using T = float3; static constexpr int N = 3; };
template <> struct WVec<32> { using T = float2; static constexpr int N = 2; };
template <> struct WVec<16> { using T = float;  static constexpr int N = 1; };
__device__ __forceinline__ float wcomp(const float4 &v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : (i == 2 ? v.z : v.w)); }
__device__ __forceinline__ float wcomp(const float3 &v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : v.z); }
__device__ __forceinline__ float wcomp(const float2 &v, int i) { return i == 0 ? v.x : v.y; }
__device__ __forceinline__ float wcomp(const float &v, int) { return v; }

template <int AW, int BW>
__device__ __forceinline__ void wgrad_block(const Wgrad2Args &w, int a0, int b0, int beg, int end,
                                            float *out, int lane) {
  using AV = typename WVec<AW>::T;
  using BV = typename WVec<BW>::T;
  constexpr int NA = WVec<AW>::N, NB = WVec<BW>::N;
  const int g = lane >> 4, l15 = lane & 15;
  // per-lane channel offsets, clamped inside the row; out-of-range channels are zeroed by select
  const int ac = a0 + NA * l15, bc = b0 + NB * l15;
  const bool aok = ac + NA <= w.ca, bok = bc + NB <= w.cb;
  const int acl = aok ? ac : 0, bcl = bok ? bc : 0;
  f32x4 acc[NA][NB];
#pragma unroll
  for (int f = 0; f < NA; ++f)
#pragma unroll
    for (int h = 0; h < NB; ++h) acc[f][h] = (f32x4){0, 0, 0, 0};

  struct Batch { AV a[4]; BV b[4]; };  // 16 pairs = 4 k-groups of 4 pairs
  // lane l15 fetches pair p0+l15 (clamped to the chunk); the pair indices run ONE batch ahead of the row loads, so
  // the dependent chain (pair -> row address -> row) never sits inside one pipeline stage
  auto load_pairs = [&](int p0) {
    int pi = p0 + l15;
    pi = pi < end ? pi : end - 1;
    return reinterpret_cast<const int2 *>(w.pairs)[pi];
  };
  auto load_rows = [&](Batch &bt, const int2 pr) {  // k-group j uses the pair held by lane 4j+g
    const int ia = w.a_col ? pr.y : pr.x, ib = w.a_col ? pr.x : pr.y;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int ra = __shfl(ia, 4 * j + g, 64), rb = __shfl(ib, 4 * j + g, 64);
      bt.a[j] = *reinterpret_cast<const AV *>(w.fa + (int64_t)ra * w.ca + acl);
      bt.b[j] = *reinterpret_cast<const BV *>(w.fb + (int64_t)rb * w.cb + bcl);
    }
  };
  auto mfma_batch = [&](const Batch &bt, int p0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool pv = aok && (p0 + 4 * j + g) < end;  // tail pairs / padded channels contribute 0
#pragma unroll
      for (int f = 0; f < NA; ++f) {
        const float av = pv ? wcomp(bt.a[j], f) : 0.f;
#pragma unroll
        for (int h = 0; h < NB; ++h)
          acc[f][h] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, wcomp(bt.b[j], h), acc[f][h], 0, 0, 0);
      }
    }
  };
  Batch b0s, b1s;
  load_rows(b0s, load_pairs(beg));
  int2 prn = load_pairs(beg + 16 < end ? beg + 16 : beg);  // pairs of the batch after the one in flight
  for (int p0 = beg; p0 < end; p0 += 32) {
    const int p2 = p0 + 32 < end ? p0 + 32 : p0;  // clamped: a redundant batch is masked out in mfma_batch
    const int p3 = p0 + 48 < end ? p0 + 48 : p0;
    load_rows(b1s, prn);
    prn = load_pairs(p2);
    __builtin_amdgcn_sched_barrier(0);
    mfma_batch(b0s, p0);
    __builtin_amdgcn_sched_barrier(0);
    load_rows(b0s, prn);
    prn = load_pairs(p3);
    __builtin_amdgcn_sched_barrier(0);
    if (p0 + 16 < end) mfma_batch(b1s, p0 + 16);  // wave-uniform
    __builtin_amdgcn_sched_barrier(0);
  }
  // tile (f,h), register r: row a0 + 4*(4g+r) + f, column b0 + 4*l15 + h  -> NB-wide stores
#pragma unroll
  for (int f = 0; f < NA; ++f) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = a0 + NA * (4 * g + r) + f;
      if (row < w.ca && bok) {
        float *d = out + (int64_t)row * w.cb + bc;
#pragma unroll
        for (int h = 0; h < NB; ++h) d[h] = acc[f][h][r];
      }
    }
  }
}

__global__ void __launch_bounds__(256, 3) wgrad2_kernel(Wgrad2Args w) {
  __shared__ int sh[3];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  if (tid == 0) find_split(w.koff, w.K, w.pch, blockIdx.x, &sh[0], &sh[1], &sh[2]);
  __syncthreads();
  const int beg = sh[1], end = sh[2];
  const int blk = blockIdx.y * 4 + wid;  // output block of this wave
  const int nag = wg_ngroups(w.ca);
  if (beg >= end || blk >= nag * w.nbg) return;
  const int ag = blk / w.nbg, bg = blk - ag * w.nbg;
  const int aw = wg_gwidth(w.ca), bw = wg_gwidth(w.cb);
  float *out = w.partial + (int64_t)blockIdx.x * w.ca * w.cb;
  const int a0 = aw * ag, b0 = bw * bg;
#define PCS_WG_CASE(A, B) if (aw == A && bw == B) { wgrad_block<A, B>(w, a0, b0, beg, end, out, lane); return; }
  PCS_WG_CASE(64, 64) PCS_WG_CASE(64, 48) PCS_WG_CASE(64, 32) PCS_WG_CASE(64, 16)
  PCS_WG_CASE(48, 64) PCS_WG_CASE(48, 48) PCS_WG_CASE(48, 32) PCS_WG_CASE(48, 16)
  PCS_WG_CASE(32, 64) PCS_WG_CASE(32, 48) PCS_WG_CASE(32, 32) PCS_WG_CASE(32, 16)
  PCS_WG_CASE(16, 64) PCS_WG_CASE(16, 48) PCS_WG_CASE(16, 32) PCS_WG_CASE(16, 16)
#undef PCS_WG_CASE
}

int wgrad_plan(const int32_t *koff_host, int K, int ca, int cb, int *pch_out) {
  // ~3072 workgroups in total (each = 4 output blocks of one split), >= 64 pairs per split
  const int nbq = (wg_ngroups(ca) * wg_ngroups(cb) + 3) / 4;
  int64_t P = koff_host[K] - koff_host[0];
  static const int total = getenv("PCS_WGRAD_WGS") ? atoi(getenv("PCS_WGRAD_WGS")) : 3072;
  int64_t target = total / nbq;
  if (target < K) target = K;
  int pch = (int)ceil_div(P > 0 ? P : 1, target);
  pch = (int)(ceil_div(pch, 32) * 32);
  if (pch < 64) pch = 64;
  int64_t ns = 0;
  for (int k = 0; k < K; ++k) ns += ceil_div((int64_t)koff_host[k + 1] - koff_host[k], pch);
  *pch_out = pch;
  return (int)ns;
}

int conv_nctt(int cout) {  // 16-column MFMA tiles per column tile: 1, 2, 3, 4, 6 or 8
  int nctt = (cout + 15) / 16;
  if (nctt > 8) nctt = 8;
  if (nctt == 5) nctt = 6;
  if (nctt == 7) nctt = 8;
  return nctt;
}
// v5 serves 16-byte-granular shapes with a contraction of at least two 32-channel steps and an even column tile
bool conv5_applies(int cin, int cout, int K) {
  return cin % 32 == 0 && cin >= 64 && cout % 4 == 0 && K <= 32 && conv_nctt(cout) % 2 == 0;
}
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

}  // namespace

#if PCS_TRACE
// debug builds only: phase timers of the last v5 launch, [block < 8192][wave < 8][8] int64 (synchronises the device)
extern "C" int pcs_debug_conv_trace(long long *host_out) {
  if (!g_trace_host_ptr || !host_out) return PCS_EINVAL;
  if (hipDeviceSynchronize() != hipSuccess) return PCS_ELAUNCH;
  return hipMemcpy(host_out, g_trace_host_ptr, (size_t)kTraceBlocks * 64 * sizeof(long long), hipMemcpyDeviceToHost) == hipSuccess ? PCS_OK : PCS_ELAUNCH;
}
#endif

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
  const int nctt = conv_nctt(cout);
  const int64_t ncol = ceil_div(cout, 16 * nctt);
  const int64_t slots = (int64_t)device_cus() * 2;
  const double ppr = (double)n_pairs / (double)n_dst;
  auto cost = [&](int T) {
    const int64_t wgs = ceil_div(n_dst, T) * ncol;
    return (double)ceil_div(wgs, slots) * (T * ppr + 8.0 * K);
  };
  // many waves of workgroups and few pairs per row (strides 1/2: 4-5.5 pairs per row, 1.28-1.36x MFMA padding at
  // 128 rows): one 8-wave workgroup per CU on 256-row tiles pads 1.15-1.17x (measured +3..4.5 %)
  if (ceil_div(n_dst, 128) * ncol >= 8 * slots && ppr < 6.5 &&
      (size_t)(257 * (16 * nctt + 4)) * 4 + 1024 <= kMaxDynLds) return 256;
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
  if (tile_rows < 16 || tile_rows > 512 || tile_rows % 16) { set_error("pcs_conv_gather_gemm_f32: tile_rows must be a multiple of 16 in [16, 512]"); return PCS_EINVAL; }
  ConvArgs a;
  a.src = src; a.W = W; a.bias = bias; a.dst = dst; a.pairs = pairs; a.seg = seg;
  a.n_dst = n_dst; a.ntiles = ceil_div(n_dst, tile_rows); a.tile_rows = tile_rows;
  a.cin = cin; a.cout = cout; a.K = K; a.src_col = src_col;
  static const int xcd = getenv("PCS_CONV_XCD") ? atoi(getenv("PCS_CONV_XCD")) : 1;
  a.xcd_remap = xcd;
  const bool vec = (cin % 4 == 0) && (cout % 4 == 0) && (((uintptr_t)src | (uintptr_t)W | (uintptr_t)dst | (uintptr_t)bias) & 15) == 0;
  hipStream_t st = as_stream(stream);
  static const int use_v1 = getenv("PCS_CONV_V1") ? atoi(getenv("PCS_CONV_V1")) : 0;
  // v5 (row-block groups of 2 sharing each W operand block) where the contraction is long enough to
  // profit (cin >= 64, measured +3..6 %); PCS_CONV_V5=0 forces v4 (debug)
  static const int v5r = getenv("PCS_CONV_V5") ? atoi(getenv("PCS_CONV_V5")) : 2;
  if (vec && !use_v1 && v5r > 0 && conv5_applies(cin, cout, K)) {
    const int nctt = conv_nctt(cout);
    a.ncoltiles = (int)ceil_div(cout, 16 * nctt);
#define PCS_CONV5_CASE(N)                                                                           \
  case N:                                                                                           \
    if (v5r == 3) return launch_conv5<N, 4, 2, 3>(a, st);                                           \
    if (v5r == 4) return launch_conv5<N, 4, 2, 4>(a, st);                                           \
    if (a.tile_rows > 160) return launch_conv5<N, 8, 2, 2>(a, st); /* one 8-wave workgroup per CU */ \
    return launch_conv5<N, 4, 2, 2>(a, st);
    switch (nctt) {
      PCS_CONV5_CASE(2)
      PCS_CONV5_CASE(4)
      PCS_CONV5_CASE(6)
      PCS_CONV5_CASE(8)
    }
#undef PCS_CONV5_CASE
  }
  if (tile_rows != 64 && tile_rows != 128) { set_error("pcs_conv_gather_gemm_f32: this shape takes tile_rows 64 or 128"); return PCS_EUNSUPPORTED; }
  if (vec && !use_v1 && K <= 32) {
    const int nctt = conv_nctt(cout);
    a.ncoltiles = (int)ceil_div(cout, 16 * nctt);
#define PCS_CONV4_CASE(N)                                                             \
  case N:                                                                             \
    return tile_rows == 128 ? launch_conv4<N, 128>(a, st) : launch_conv4<N, 64>(a, st);
    switch (nctt) {
      PCS_CONV4_CASE(1)
      PCS_CONV4_CASE(2)
      PCS_CONV4_CASE(3)
      PCS_CONV4_CASE(4)
      PCS_CONV4_CASE(6)
      PCS_CONV4_CASE(8)
    }
#undef PCS_CONV4_CASE
  }
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
  const int ns = wgrad_plan(koff_host, K, ca, cb, &pch);
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
  const int ns = wgrad_plan(koff_host, K, ca, cb, &pch);
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
  static const int use_v1 = getenv("PCS_WGRAD_V1") ? atoi(getenv("PCS_WGRAD_V1")) : 0;
  if (vec && !use_v1) {
    Wgrad2Args w2;
    w2.fa = fa; w2.fb = fb; w2.pairs = pairs; w2.koff = koff_dev; w2.partial = reinterpret_cast<float *>(ws);
    w2.ca = ca; w2.cb = cb; w2.K = K; w2.a_col = a_col; w2.pch = pch; w2.nbg = wg_ngroups(cb);
    const int nblk = wg_ngroups(ca) * wg_ngroups(cb);
    hipLaunchKernelGGL(wgrad2_kernel, dim3((unsigned)ns, (unsigned)ceil_div(nblk, 4)), dim3(256), 0, st, w2);
  } else {
    dim3 grid((unsigned)ns, (unsigned)ceil_div(ca, 128), (unsigned)ceil_div(cb, 128));
    if (vec) hipLaunchKernelGGL(wgrad_kernel<true>, grid, dim3(256), 0, st, w);
    else hipLaunchKernelGGL(wgrad_kernel<false>, grid, dim3(256), 0, st, w);
  }
  int rc = check_launch("pcs_conv_wgrad_f32");
  if (rc) return rc;
  if (vec && (((uintptr_t)ws | (uintptr_t)gW) & 15) == 0) {
    hipLaunchKernelGGL(wgrad_reduce4_kernel, dim3((unsigned)ceil_div(cc, 64), K), dim3(256), 0, st,
                       reinterpret_cast<const float *>(ws), koff_dev, (int)K, pch, cc, gW);
  } else {
    int gx = (int)ceil_div(cc, 256);
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(gx, K), dim3(256), 0, st,
                       reinterpret_cast<const float *>(ws), koff_dev, (int)K, pch, cc, gW);
  }
  return check_launch("pcs_conv_wgrad_f32(reduce)");
}
