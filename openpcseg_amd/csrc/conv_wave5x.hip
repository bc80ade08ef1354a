// fp32 fused convolution on the 16-bit MFMAs ("bf16x3"): fp32 features and fp32 weights, every value split into three
// bf16 planes (hi, mid, lo; round-to-nearest-even at every step: hi + mid + lo is exact to 2^-26 of the value), six of the
// nine plane products accumulated in fp32 (lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi: what is dropped is below 2^-24
// of a product, the size of the rounding of an fp32 FMA). gfx950 has no TF32 / xf32 path and runs v_mfma_f32_16x16x4_f32
// at 1/16 of the bf16 rate, so six 16x16x32 bf16 MFMAs replace eight 16x16x4 fp32 ones at a third of their pipe time.
// Replaces the same reference dataflow as conv_wave5.hip (TS:torchsparse/backend/convolution/convolution_cuda.cu:53-165,
// gather -> mm -> scatter in fp32); it is OPT-IN (functional.set_conv_policy("bf16x3")): the library's fp32 default stays on
// fp32 MFMA arithmetic. Results are fp32-grade, not bit-identical to an fp32 FMA chain (tests/test_dense_parity.py holds its
// error against float64 to at most twice the fp32 MFMA kernel's).
//
// Structure = conv_wave5h.hip (output-stationary tile in LDS, wave-autonomous row-block groups, ticket-ordered phased commit,
// shared epilogue with BatchNorm partials), operand side:
//   * A: lane (n = lane & 15, g = lane >> 4) gathers the 8 floats src[row_n][32 s + 8 g .. +7] (two 16-byte loads) and splits
//     them in registers (v_cvt_pk_bf16_f32 on pairs) right before the MFMAs of the step;
//   * B: the weights are split and re-packed once per layer call (pcs_conv_prepare_weights_x3) into three planes of MFMA
//     fragment order, plane p at Wp + p * plane_bytes, each laid out like the half kernel's prepared weights;
//   * B fragments are SINGLE-buffered and rolling: the fragments of a pair of 16-column tiles are re-loaded for the next step
//     right after that pair's 6 R MFMAs each have been issued -- with six MFMAs per product the rest of the step (>= 1000
//     cycles) covers the L2 latency, and three planes of B double-buffered would not fit the registers;
//   * column tiles are at most 96 wide: >= 112 output columns run as 64-column tiles (as the fp32 kernel does);
//   * groups hold R = 4 row blocks at 64 columns, 3 at 96, 2 at 32: the kernel is bound by the vector-memory address path
//     (TA 78-87 % busy, MFMA pipe 29-37 %; without the MFMAs the launch takes the same time), a column tile costs 2 A loads
//     + 3 N / R weight-fragment loads per row block and step, and R is what 256 registers allow (profiles/round3_convx.md).
#include "conv_common.h"

#ifndef PCS_ABLATEX
#define PCS_ABLATEX 0  /* debug builds: 1 no operand split (planes = raw bits), 2 no B reloads inside a group, 3 no A reloads, 4 no MFMA */
#endif

using namespace pcs;

namespace {

typedef __bf16 x_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 x_bf16x2 __attribute__((ext_vector_type(2)));
typedef float x_f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x4 mfma_x(const uint4 &a, const uint4 &b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(x_bf16x8, a), __builtin_bit_cast(x_bf16x8, b), c, 0, 0, 0);
}

// (x, y) fp32 -> the packed bf16 pairs of the three planes (round to nearest even at every step)
__device__ __forceinline__ void split3(float x, float y, uint32_t &hi, uint32_t &mid, uint32_t &lo) {
  const x_f32x2 v = {x, y};
  const x_bf16x2 h = __builtin_convertvector(v, x_bf16x2);
  x_f32x2 r1 = v - __builtin_convertvector(h, x_f32x2);
  // +-Inf, and finite values above the bf16 maximum that round to Inf: hi carries the Inf and the lower planes are zero
  // (Inf - Inf would make them NaN). The products of that Inf with the signed lower planes of the other operand can still
  // sum to NaN: an output the fp32 kernel makes Inf is non-finite here, Inf or NaN. NaN stays NaN through hi.
  if (__builtin_isinf((float)h.x)) r1.x = 0.f;
  if (__builtin_isinf((float)h.y)) r1.y = 0.f;
  const x_bf16x2 m = __builtin_convertvector(r1, x_bf16x2);
  const x_f32x2 r2 = r1 - __builtin_convertvector(m, x_f32x2);
  const x_bf16x2 l = __builtin_convertvector(r2, x_bf16x2);
  hi = __builtin_bit_cast(uint32_t, h); mid = __builtin_bit_cast(uint32_t, m); lo = __builtin_bit_cast(uint32_t, l);
}

// 16-column MFMA tiles per column tile: 2, 4 or 6 (>= 112 output columns as 64-column tiles)
__host__ __device__ inline int convx_nctt(int cout) {
  const int n = (cout + 15) / 16;
  if (n <= 2) return 2;
  if (n <= 4) return 4;
  if (n <= 6) return 6;
  return 4;
}

// local column of lane n of 16-column tile tl inside a column tile (the 4- / 2-interleave the commit and epilogue assume)
__host__ __device__ inline int x_local_col(int nctt, int tl, int n) {
  const int n4 = nctt / 4;
  if (tl < 4 * n4) return 64 * (tl / 4) + 4 * n + (tl % 4);
  return 64 * n4 + 2 * n + (tl - 4 * n4);
}

// Wp plane p, block (k, global 16-column tile gt, step s) = 64 lanes x 8 bf16; lane 16 g + n, element j =
//   plane_p(Wmath[k][32 s + 8 g + j][column(gt, n)]),  Wmath[k][c][col] = transpose ? W[k][col][c] : W[k][c][col].
// Columns >= ccols and channels >= ccon are zero in every plane.
__global__ void __launch_bounds__(256) prepare_weights_x3_kernel(const float *__restrict__ W, int K, int A, int B, int transpose,
                                                                 int nctt, int nt16, int ns, uint4 *__restrict__ Wp) {
  const int ccon = transpose ? B : A, ccols = transpose ? A : B;
  const int64_t total = (int64_t)K * nt16 * ns * 64;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int lane = (int)(i & 63);
    int64_t b = i >> 6;
    const int s = (int)(b % ns); b /= ns;
    const int gt = (int)(b % nt16);
    const int k = (int)(b / nt16);
    const int n = lane & 15, g = lane >> 4;
    const int col = (gt / nctt) * 16 * nctt + x_local_col(nctt, gt % nctt, n);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = 32 * s + 8 * g + j;
      v[j] = 0.f;
      if (col < ccols && c < ccon)
        v[j] = transpose ? W[((int64_t)k * A + col) * B + c] : W[((int64_t)k * A + c) * B + col];
    }
    uint4 hi, mid, lo;
    split3(v[0], v[1], hi.x, mid.x, lo.x); split3(v[2], v[3], hi.y, mid.y, lo.y);
    split3(v[4], v[5], hi.z, mid.z, lo.z); split3(v[6], v[7], hi.w, mid.w, lo.w);
    Wp[i] = hi; Wp[total + i] = mid; Wp[2 * total + i] = lo;
  }
}

struct ConvArgsX {
  const char *src;    // (n_src, cin) fp32
  const char *Wp;     // prepared weights: three planes, fragment order
  const float *bias;  // fp32, may be NULL
  float *dst;         // (n_dst, cout) fp32
  const int32_t *pairs;
  const int32_t *seg;
  int64_t n_dst;
  int64_t ntiles;
  int64_t plane_bytes;
  int cin, cout, K, src_col, ncoltiles, tile_rows, nt16, ns;
  double *stats;         // optional [ntiles][2][cout], as ConvArgs::stats
  const int32_t *order;  // optional [ntiles]: workgroup slot -> row tile (heaviest first), as ConvArgs::order
};

template <int NCTT, int NW_, int R_>
struct Conv5xCfg {
  static constexpr int NW = NW_;
  static constexpr int R = R_;
  static constexpr int NT = 64 * NW;
  static constexpr int CT = 16 * NCTT;
  static constexpr int ACS = CT + 4;
  static constexpr int N4 = NCTT / 4;
  static constexpr int N2 = (NCTT % 4) / 2;
  static constexpr size_t lds_bytes(int T) { return (size_t)((T + 1) * ACS) * 4 + 5 * 33 * 4 + 16; }
};

template <int NCTT, int NW, int MINW, int R, bool TAIL>
__global__ void __launch_bounds__(64 * NW, MINW) conv_os5x_kernel(ConvArgsX a) {
  using C = Conv5xCfg<NCTT, NW, R>;
  static_assert(NCTT % 2 == 0 && R >= 2 && R <= 4, "tile pairs; partial groups hold 1 .. R - 1 row blocks");
  const int T = a.tile_rows;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float *acc_l = reinterpret_cast<float *>(smem);            // [T+1][ACS], row T = sink for padding rows
  int *kl_k = reinterpret_cast<int *>(acc_l + (T + 1) * C::ACS);  // [32] offset id
  int *kl_s = kl_k + 32;                                     // [32] first pair
  int *kl_m = kl_s + 32;                                     // [32] #pairs
  int *kl_g = kl_m + 32;                                     // [33] first FULL group (prefix over the offsets)
  int *kl_h = kl_g + 33;                                     // [33] first partial group (prefix)
  int *commit = kl_h + 33;
  const unsigned commit_lds = (unsigned)(size_t)(__attribute__((address_space(3))) int *)commit;  // LDS byte addresses
  const unsigned acc_lds = (unsigned)(size_t)(__attribute__((address_space(3))) float *)acc_l;
  __shared__ int nk_s;

  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, l15 = lane & 15;
  const unsigned bid = blockIdx.x;
  int64_t slot = bid / a.ncoltiles;
  int ctile = bid % a.ncoltiles;
  if (a.order) {  // tiles dealt round-robin over the XCDs, the column tiles of one row tile back to back on one XCD
    const unsigned xcd = bid & 7, idx = bid >> 3;
    slot = (int64_t)(idx / a.ncoltiles) * 8 + xcd;
    ctile = idx % a.ncoltiles;
    if (slot >= a.ntiles) return;  // the grid is padded to 8 * ncoltiles
  }
  const int64_t tile = a.order ? (int64_t)a.order[slot] : slot;
  const int n0 = ctile * C::CT;
  const int64_t row0 = tile * T;
  const int64_t nt1 = a.ntiles + 1;

  if (wid == 0) {  // non-empty offsets of this tile + prefix of their row-block groups (as conv_wave5.hip)
    const int k = lane;
    int s0 = 0, m = 0;
    if (k < a.K) {
      s0 = a.seg[(int64_t)k * nt1 + tile];
      m = a.seg[(int64_t)k * nt1 + tile + 1] - s0;
    }
    const unsigned long long mask = __ballot(m > 0);
    const int nrb = (m + 15) >> 4;
    const int nfull = nrb / R, npart = (nrb % R) ? 1 : 0;
    int incl = nfull | (npart << 16);
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
  {  // zero the tile: (T + 1) * ACS floats, a multiple of four
    float4 *z = reinterpret_cast<float4 *>(acc_l);
    const int n4 = (T + 1) * (C::ACS / 4);
    for (int i = tid; i < n4; i += C::NT) z[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  __syncthreads();
  const int nk = __builtin_amdgcn_readfirstlane(nk_s);
  const int total_full = nk > 0 ? __builtin_amdgcn_readfirstlane(kl_g[nk]) : 0;
  const int total_grp = nk > 0 ? total_full + __builtin_amdgcn_readfirstlane(kl_h[nk]) : 0;

  // B fragments of this column tile: 16-column tiles that do not exist (beyond cout) read tile 0, results dropped
  const int gt0 = ctile * NCTT;
  const int NS = a.ns;
  int boff[NCTT];  // byte offset of tile t's fragment blocks relative to a group's Wk
#pragma unroll
  for (int t = 0; t < NCTT; ++t) boff[t] = ((gt0 + t < a.nt16) ? t : 0) * NS * 1024;
  const int64_t PB = a.plane_bytes;
  // TAIL: channels 32 (NS-1) + 8 g .. +7 of the last step exist only below cin; lane groups beyond it step back to the
  // row's last 8 channels (tail_back bytes) and contribute zeros
  const int tail_over = TAIL ? 32 * (NS - 1) + 8 * g + 8 - a.cin : 0;
  const bool tail_ok = tail_over <= 0;
  const int tail_back = tail_ok ? 0 : 4 * tail_over;

  struct Ctx {  // one group: R row blocks of one offset
    const char *srow[R];  // this lane's 32 bytes of step 0 of the row it gathers
    const char *Wk;       // plane 0, fragment blocks of (offset, first 16-column tile of this column tile), this lane's 16 bytes
    int dloc[R];
    int nr;
    unsigned vmask;
  };
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
      cx.srow[r] = a.src + ((int64_t)(a.src_col ? pr[r].y : pr[r].x) * a.cin + 8 * g) * 4;
      cx.dloc[r] = ((vmask >> r) & 1u) ? (int)((a.src_col ? pr[r].x : pr[r].y) - row0) : T;
    }
    cx.vmask = vmask;
    cx.nr = nr;
    cx.Wk = a.Wp + (((int64_t)kl_k[i_k & 31] * a.nt16 + gt0) * NS) * 1024 + lane * 16;
  };
  // operand registers that live across groups: the raw A pieces and the three B planes of the step about to be computed
  uint4 araw[R][2];
  uint4 bfr[3][NCTT];
  auto load_a = [&](const Ctx &cx, int s) {
    const int aoff = TAIL ? s * 128 - (s == NS - 1 ? tail_back : 0) : s * 128;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      araw[r][0] = *reinterpret_cast<const uint4 *>(cx.srow[r] + aoff);
      araw[r][1] = *reinterpret_cast<const uint4 *>(cx.srow[r] + aoff + 16);
    }
  };
  auto load_b = [&](const char *wk_s, int t) {  // wk_s = Wk + s * 1024 of the step to load
#pragma unroll
    for (int p = 0; p < 3; ++p) bfr[p][t] = *reinterpret_cast<const uint4 *>(wk_s + p * PB + boff[t]);
  };

  int i = 0;
  Ctx cur;
  if (wid < total_grp) {
    int pidx[R]; unsigned vm; int nr;
    locate(wid, i, pidx, vm, nr);
    int2 pr[R];
#pragma unroll
    for (int r = 0; r < R; ++r) pr[r] = reinterpret_cast<const int2 *>(a.pairs)[pidx[r]];
    make_ctx(cur, pr, vm, nr, i);
    load_a(cur, 0);
#pragma unroll
    for (int t = 0; t < NCTT; ++t) load_b(cur.Wk, t);
  }
  // one straight-line body per group loop (full groups: R row blocks, partial groups: one), see conv_wave5.hip
  auto run_group = [&](const int grp, auto nrc_tag) {
    constexpr int NRC = decltype(nrc_tag)::value;
    const int grpn = grp + C::NW < total_grp ? grp + C::NW : grp;
    int in = i, pidx_n[R], nr_n; unsigned vm_n;
    locate(grpn, in, pidx_n, vm_n, nr_n);
    int2 pr_n[R];
#pragma unroll
    for (int r = 0; r < R; ++r) pr_n[r] = reinterpret_cast<const int2 *>(a.pairs)[pidx_n[r]];
    Ctx nxt;
    make_ctx(nxt, pr_n, vm_n, nr_n, in);

    f32x4 acc[R][NCTT];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int t = 0; t < NCTT; ++t) acc[r][t] = (f32x4){0, 0, 0, 0};
    const unsigned vmask = cur.vmask;
    for (int s = 0; s < NS; ++s) {  // wave-uniform
      // split this step's A pieces into planes; rows that are padding (or, TAIL, lane groups beyond cin) contribute zeros
      const bool lane_ok = !TAIL || s != NS - 1 || tail_ok;
      uint4 ap[R][3];
#pragma unroll
      for (int r = 0; r < NRC; ++r) {
        const bool ok = ((vmask >> r) & 1u) && lane_ok;
        uint4 x0 = araw[r][0], x1 = araw[r][1];
        if (!ok) { x0 = make_uint4(0u, 0u, 0u, 0u); x1 = x0; }
#if PCS_ABLATEX == 1
        ap[r][0] = x0; ap[r][1] = x1; ap[r][2] = make_uint4(x0.x ^ x1.y, x0.y, x1.z, x0.w);
#else
        split3(__uint_as_float(x0.x), __uint_as_float(x0.y), ap[r][0].x, ap[r][1].x, ap[r][2].x);
        split3(__uint_as_float(x0.z), __uint_as_float(x0.w), ap[r][0].y, ap[r][1].y, ap[r][2].y);
        split3(__uint_as_float(x1.x), __uint_as_float(x1.y), ap[r][0].z, ap[r][1].z, ap[r][2].z);
        split3(__uint_as_float(x1.z), __uint_as_float(x1.w), ap[r][0].w, ap[r][1].w, ap[r][2].w);
#endif
      }
      // the operands of the NEXT step: this group's step s + 1, or step 0 of the wave's next group
      const bool more = s + 1 < NS;
      const Ctx &cn = more ? cur : nxt;
      const int sn = more ? s + 1 : 0;
#if PCS_ABLATEX == 3
      if (!more)
#endif
      load_a(cn, sn);
      const char *wk_n = cn.Wk + (int64_t)sn * 1024;
#pragma unroll
      for (int t = 0; t < NCTT; t += 2) {
        // six plane products, small terms first; consecutive MFMAs go to different accumulators (2 tiles x NRC row blocks)
#define PCS_X3_PROD(PA, PB_)                                                                        \
  _Pragma("unroll") for (int tt = 0; tt < 2; ++tt)                                                  \
  _Pragma("unroll") for (int r = 0; r < NRC; ++r)                                                    \
      acc[r][t + tt] = PCS_ABLATEX == 4 ? (f32x4){acc[r][t + tt][0] + __uint_as_float(ap[r][PA].x ^ bfr[PB_][t + tt].x), 0, 0, 0} \
                                         : mfma_x(ap[r][PA], bfr[PB_][t + tt], acc[r][t + tt]);
        PCS_X3_PROD(2, 0) PCS_X3_PROD(0, 2) PCS_X3_PROD(1, 1) PCS_X3_PROD(1, 0) PCS_X3_PROD(0, 1) PCS_X3_PROD(0, 0)
#undef PCS_X3_PROD
#if PCS_ABLATEX == 2
        if (!more)
#endif
        {
          load_b(wk_n, t);      // rolling single buffer: behind the last MFMA that reads these fragments
          load_b(wk_n, t + 1);
        }
      }
    }
    // ---- in-order commit of the group's row blocks (conv_wave5.hip: addresses before the ticket wait, three fenced phases,
    // bare ds_write_b32 ticket behind the tile writes) -----------------------------------------------------------------
    unsigned dq[R][4], dp[R][4];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const unsigned dl = (unsigned)__shfl(cur.dloc[r], 4 * g + j, 64);
        dq[r][j] = acc_lds + 4u * dl * C::ACS + 16u * l15;
        dp[r][j] = acc_lds + 4u * dl * C::ACS + 256u * C::N4 + 8u * l15;
        asm volatile("" : "+v"(dq[r][j]), "+v"(dp[r][j]));
      }
    if (lane == 0) {
      while (__hip_atomic_load(commit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != grp)
        __builtin_amdgcn_s_sleep(1);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    __builtin_amdgcn_s_setprio(3);
    {
      typedef float v2f __attribute__((ext_vector_type(2)));
      typedef __attribute__((address_space(3))) const f32x4 lds_cf4;
      typedef __attribute__((address_space(3))) const v2f lds_cf2;
      typedef __attribute__((address_space(3))) f32x4 lds_f4;
      typedef __attribute__((address_space(3))) v2f lds_f2;
#pragma unroll
      for (int r = 0; r < NRC; ++r) {  // one round per row block (registers)
        f32x4 v4[4][C::N4 > 0 ? C::N4 : 1];
        v2f v2[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
          for (int q = 0; q < C::N4; ++q) v4[j][q] = *(lds_cf4 *)(size_t)(dq[r][j] + 256u * q);
          if (C::N2) v2[j] = *(lds_cf2 *)(size_t)dp[r][j];
        }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
          for (int q = 0; q < C::N4; ++q) {
            v4[j][q].x += acc[r][4 * q + 0][j]; v4[j][q].y += acc[r][4 * q + 1][j];
            v4[j][q].z += acc[r][4 * q + 2][j]; v4[j][q].w += acc[r][4 * q + 3][j];
          }
          if (C::N2) { v2[j].x += acc[r][4 * C::N4 + 0][j]; v2[j].y += acc[r][4 * C::N4 + 1][j]; }
        }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
          for (int q = 0; q < C::N4; ++q) *(lds_f4 *)(size_t)(dq[r][j] + 256u * q) = v4[j][q];
          if (C::N2) *(lds_f2 *)(size_t)dp[r][j] = v2[j];
        }
        asm volatile("" ::: "memory");
      }
    }
#if PCS_COMMIT_NOWAIT
    // as conv_wave5.hip: the ticket store stays behind the tile writes in program order and the LDS executes one wave's
    // instructions in order (the hardware assumption is stated once); PCS_COMMIT_NOWAIT=0 is the
    // fenced form, bit-identical (tests/test_dense_parity.py::test_commit_variants_bit_identical on a variant library)
    if (lane == 0) asm volatile("ds_write_b32 %0, %1" ::"v"(commit_lds), "v"(grp + 1) : "memory");
#else
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    if (lane == 0) __hip_atomic_store(commit, grp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
    __builtin_amdgcn_s_setprio(0);
    cur = nxt;
    i = in;
  };
  {  // wave-uniform loops, no barrier inside: the full groups, then the partial ones (= the commit order)
    int grp = wid;
    for (; grp < total_full; grp += C::NW) run_group(grp, std::integral_constant<int, R>{});
    for (; grp < total_grp; grp += C::NW) {
      const int nrp = __builtin_amdgcn_readfirstlane(cur.nr);  // wave-uniform: 1 .. R - 1 row blocks
      if (R >= 4 && nrp == 3) run_group(grp, std::integral_constant<int, (R >= 4 ? 3 : 1)>{});
      else if (R >= 3 && nrp == 2) run_group(grp, std::integral_constant<int, (R >= 3 ? 2 : 1)>{});
      else run_group(grp, std::integral_constant<int, 1>{});
    }
  }
  __syncthreads();
  const int rows = (int)((a.n_dst - row0) < (int64_t)T ? (a.n_dst - row0) : (int64_t)T);
  float *drow = a.dst + row0 * a.cout + n0;
  const int ldd = a.cout;
  conv_tile_epilogue<C::CT, C::NT>(acc_l, C::ACS, rows, n0, a.cout, a.bias, a.stats ? a.stats + tile * 2 * a.cout : nullptr, tid,
                                   [&](int r, int cq, const float4 &v) {
                                     *reinterpret_cast<float4 *>(drow + (int64_t)r * ldd + cq) = v;
                                     return v;
                                   });
}

template <int NCTT, int NW, int MINW, int R, bool TAIL>
int launch_conv5x(const ConvArgsX &a, hipStream_t st) {
  using C = Conv5xCfg<NCTT, NW, R>;
  const int64_t nblocks = a.order ? ceil_div(a.ntiles, 8) * 8 * a.ncoltiles : a.ntiles * a.ncoltiles;
  if (nblocks <= 0) return PCS_OK;
  if (nblocks > 0x7FFFFFFF) { set_error("pcs_conv_x3: grid too large"); return PCS_EUNSUPPORTED; }
  auto kern = conv_os5x_kernel<NCTT, NW, MINW, R, TAIL>;
  const size_t lds = C::lds_bytes(a.tile_rows);
  if (lds > kMaxDynLds) { set_error("pcs_conv_x3: tile_rows too large for this column tile"); return PCS_EUNSUPPORTED; }
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxDynLds);
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)nblocks), dim3(C::NT), lds, st, a);
  return check_launch("pcs_conv_gather_gemm_f32_bf16x3(wave5x)");
}

inline bool convx_applies(int cin, int cout, int K) {
  return cin % 8 == 0 && cin >= 32 && cout % 4 == 0 && cout >= 32 && K <= 32 && convx_nctt(cout) * 16 >= 32;
}

}  // namespace

extern "C" int pcs_conv_x3_applies(int32_t cin, int32_t cout, int32_t K) { return convx_applies(cin, cout, K) ? 1 : 0; }

extern "C" int32_t pcs_conv_x3_column_tiles(int32_t cout) { return cout > 0 ? convx_nctt(cout) : 0; }

extern "C" int32_t pcs_conv_x3_emits_bn_partials(int32_t cin, int32_t cout, int32_t K, int32_t tile_rows) {
  if (!convx_applies(cin, cout, K) || tile_rows < 16) return 0;
  const int nctt = convx_nctt(cout);
  const size_t lds = (size_t)((tile_rows + 1) * (16 * nctt + 4)) * 4 + 1024;
  return conv_stats_fit(tile_rows, 16 * nctt, 2 * lds > 160 * 1024 ? 512 : 256) ? 1 : 0;
}

extern "C" size_t pcs_conv_prepared_weights_x3_bytes(int32_t K, int32_t ccon, int32_t ccols) {
  if (K <= 0 || ccon <= 0 || ccols <= 0 || ccon % 8) return 0;
  const int nctt = convx_nctt(ccols);
  return 3 * (size_t)K * (size_t)ceil_div(ccols, 16 * nctt) * nctt * (size_t)ceil_div(ccon, 32) * 1024;
}

extern "C" int pcs_conv_prepare_weights_x3(const float *W, int32_t K, int32_t A, int32_t B, int32_t transpose, void *Wp,
                                           void *stream) {
  const int ccon = transpose ? B : A, ccols = transpose ? A : B;
  if (K <= 0 || A <= 0 || B <= 0 || !W || !Wp || !convx_applies(ccon, ccols, K)) {
    set_error("pcs_conv_prepare_weights_x3: bad args / shape not served by the bf16x3 kernel");
    return PCS_EINVAL;
  }
  const int nctt = convx_nctt(ccols), nt16 = (int)ceil_div(ccols, 16 * nctt) * nctt, ns = (int)ceil_div(ccon, 32);
  const int64_t total = (int64_t)K * nt16 * ns * 64;
  hipLaunchKernelGGL(prepare_weights_x3_kernel, dim3(stream_grid(total, 256)), dim3(256), 0, as_stream(stream), W, K, A, B,
                     transpose, nctt, nt16, ns, reinterpret_cast<uint4 *>(Wp));
  return check_launch("pcs_conv_prepare_weights_x3");
}

extern "C" int pcs_conv_gather_gemm_f32_bf16x3(const float *src, int64_t n_src, int32_t cin, const void *Wp, int32_t K,
                                               int32_t cout, const int32_t *pairs, int32_t src_col, const int32_t *seg,
                                               int32_t tile_rows, int64_t n_dst, const float *bias, float *dst,
                                               double *bn_partial, const int32_t *tile_order, void *stream) {
  if (cin <= 0 || cout <= 0 || K <= 0 || n_dst < 0 || n_src < 0 || (src_col != 0 && src_col != 1)) {
    set_error("pcs_conv_gather_gemm_f32_bf16x3: bad sizes");
    return PCS_EINVAL;
  }
  if (!convx_applies(cin, cout, K)) {
    set_error("pcs_conv_gather_gemm_f32_bf16x3: shape not served (needs cin %% 8 == 0, cin >= 32, cout %% 4 == 0, cout >= 32, K <= 32)");
    return PCS_EUNSUPPORTED;
  }
  if (n_dst == 0) return PCS_OK;
  if (!Wp || !seg || !dst || (n_src > 0 && !src)) { set_error("pcs_conv_gather_gemm_f32_bf16x3: null pointer"); return PCS_EINVAL; }
  if (((uintptr_t)src | (uintptr_t)Wp | (uintptr_t)bias | (uintptr_t)dst) & 15) { set_error("pcs_conv_gather_gemm_f32_bf16x3: misaligned pointer"); return PCS_EINVAL; }
  if (tile_rows < 16 || tile_rows > 512 || tile_rows % 16) { set_error("pcs_conv_gather_gemm_f32_bf16x3: tile_rows must be a multiple of 16 in [16, 512]"); return PCS_EINVAL; }
  ConvArgsX a;
  a.src = reinterpret_cast<const char *>(src); a.Wp = reinterpret_cast<const char *>(Wp); a.bias = bias; a.dst = dst;
  a.pairs = pairs; a.seg = seg; a.n_dst = n_dst; a.ntiles = ceil_div(n_dst, tile_rows); a.tile_rows = tile_rows;
  a.cin = cin; a.cout = cout; a.K = K; a.src_col = src_col; a.stats = bn_partial; a.order = tile_order;
  const int nctt = convx_nctt(cout);
  a.ncoltiles = (int)ceil_div(cout, 16 * nctt);
  a.nt16 = a.ncoltiles * nctt;
  a.ns = (int)ceil_div(cin, 32);
  a.plane_bytes = (int64_t)K * a.nt16 * a.ns * 1024;
  const size_t lds = (size_t)((tile_rows + 1) * (16 * nctt + 4)) * 4 + 1024;
  if (lds > kMaxDynLds) { set_error("pcs_conv_gather_gemm_f32_bf16x3: tile_rows too large for this column tile"); return PCS_EUNSUPPORTED; }
  if (bn_partial && !conv_stats_fit(tile_rows, 16 * nctt, 2 * lds > 160 * 1024 ? 512 : 256)) {
    set_error("pcs_conv_gather_gemm_f32_bf16x3: this tile height does not produce BatchNorm partials");
    return PCS_EUNSUPPORTED;
  }
  const bool nw8 = 2 * lds > 160 * 1024;  // 4-wave workgroups while two of them fit a CU's LDS, else one 8-wave workgroup
  const bool tail = (cin % 32) != 0;
  hipStream_t st = as_stream(stream);
  // Row blocks per group. The kernel is bound by the vector-memory address path (TA 78-87 % busy, MFMA pipe 29-37 %,
  // profiles/round3_convx.md): per row block and step a column tile costs 2 A loads + 3 N / R weight-fragment loads, so R is
  // the lever -- four row blocks per group at <= 64 columns, three at 96 (what the 256 registers of two waves per SIMD hold).
  static const int force_r = getenv("PCS_CONVX_R") ? atoi(getenv("PCS_CONVX_R")) : 0;  // A/B
  // measured (profiles/round3_convx.md): R = 4 at 64 columns 1.11x R = 2 (256->256 985 -> 868 us), R = 3 at 96 columns 1.08-1.10x,
  // 32 columns lose with R > 2
  const int rsel = force_r >= 2 && force_r <= 4 ? (nctt == 6 && force_r > 3 ? 3 : force_r) : (nctt == 6 ? 3 : (nctt == 4 ? 4 : 2));
#define PCS_CONV5X_R(N, RR)                                                                         \
  if (tail) return nw8 ? launch_conv5x<N, 8, 2, RR, true>(a, st) : launch_conv5x<N, 4, 2, RR, true>(a, st);  \
  return nw8 ? launch_conv5x<N, 8, 2, RR, false>(a, st) : launch_conv5x<N, 4, 2, RR, false>(a, st);
#define PCS_CONV5X_CASE(N)                                                                          \
  case N:                                                                                           \
    if (rsel == 4 && N < 6) { PCS_CONV5X_R(N, (N < 6 ? 4 : 3)) }                                      \
    if (rsel >= 3) { PCS_CONV5X_R(N, 3) }                                                           \
    { PCS_CONV5X_R(N, 2) }
  switch (nctt) {
    PCS_CONV5X_CASE(2)
    PCS_CONV5X_CASE(4)
    PCS_CONV5X_CASE(6)
  }
#undef PCS_CONV5X_CASE
#undef PCS_CONV5X_R
  set_error("pcs_conv_gather_gemm_f32_bf16x3: unreachable");
  return PCS_EINVAL;
}
