// Generic fused convolution kernel (block-synchronous): the fallback for shapes whose rows are not 16-byte granular
// (e.g. Cin = 5) or with more than 32 offsets. Reference dataflow being replaced
// (TS:torchsparse/backend/convolution/convolution_cuda.cu): per kernel offset k a gather kernel (:14-24), a cuBLAS
// mm_out (:149) and a scatter kernel (:27-37), i.e. 3 launches and 2 extra HBM round trips of the gathered tile.
//
// One workgroup owns `T` consecutive destination rows x a 32*CG column tile and keeps the fp32 accumulator tile in LDS
// for the whole kernel. For each offset k the pairs whose destination falls in the tile form ONE contiguous rulebook
// slice (pairs are sorted by destination inside an offset), so the workgroup
//   1. reads the slice (src row, dst row) -> LDS,
//   2. gathers the m src rows (cin chunk of 32) into a compact LDS tile with coalesced 16 B loads,
//   3. stages the W[k] chunk (32 x CT) in LDS,
//   4. runs 16x16x4 fp32 MFMAs on the compact m x 32 tile (only ceil(m/16) row blocks issue),
//   5. adds the compact result rows into the accumulator tile through the dst-row map.
// Every destination row is written exactly once at the end: no atomics, no zero fill of dst.
#include "conv_common.h"

using namespace pcs;

namespace {

constexpr int CK = 32;        // cin chunk staged per step
constexpr int AS = CK + 2;    // abuf row stride (words): rows*2+g distinct mod 32 -> no conflicts

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

}  // namespace

int pcs::launch_conv_block(ConvArgs a, bool vec, hipStream_t st) {
  // column tile: 32*CG with CG in 1..4; wider outputs are covered by several column tiles
  int cg = (a.cout + 31) / 32;
  if (cg > 4) cg = 4;
  a.ncoltiles = (int)ceil_div(a.cout, 32 * cg);
#define PCS_CONV_CASE(CGv, RGv)                                                       \
  case CGv:                                                                           \
    return a.tile_rows == 128 ? launch_conv<CGv, RGv, 128>(a, vec, st)                \
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
