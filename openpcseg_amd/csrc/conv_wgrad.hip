// Weight gradient of the sparse convolution: split reduction over the rulebook with the PAIR axis as the MFMA
// contraction (fp32, v_mfma_f32_16x16x4_f32), register-only; a second kernel sums the splits in a fixed order.
#include "conv_common.h"

using namespace pcs;

namespace {

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

// The same lookup by a whole wave at once (K <= 64): lane k loads its offset's slice bounds, one wave scan finds the
// offset of `split` -- one load latency instead of up to K dependent ones at the head of every workgroup.
// `split` is the workgroup's position in LAUNCH order; `*gsid` the global split (offset-major) whose partial block it writes.
// interleave = 0: launch order = offset-major: the resident workgroups sweep one offset after the other, and each of the K
// sweeps streams the whole level's rows (both operands) from HBM -- the level (2 x 222 MB in bf16 at stride 1) does not survive
// in the 256 MB Infinity Cache from one sweep to the next.
// interleave = 1 [r6, measured neutral]: launch order = position-major. Round r (of R = the largest split count of an offset) holds, for every
// offset k, its split floor(r ns_k / R) when that value is new in round r: the resident workgroups then cover the SAME stretch of
// destination rows under all K offsets at once, so a feature row fetched for one offset serves the others out of L2 / the
// Infinity Cache. The splits, the partial blocks and the fixed-order reduction are unchanged: bit-identical results.
__device__ __forceinline__ void find_split_wave(const int32_t *koff, int K, int pch, int split, int lane, int *beg,
                                                int *end, int *gsid, int interleave) {
  if (K > 64) {
    int k;
    find_split(koff, K, pch, split, &k, beg, end);
    *gsid = split;
    return;
  }
  const int lo = lane < K ? koff[lane] : 0, hi = lane < K ? koff[lane + 1] : 0;
  const int ns = (hi - lo + pch - 1) / pch;
  int incl = ns;
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(incl, o, 64);
    if (lane >= o) incl += t;
  }
  int src, sidx;   // offset (lane) and split inside it
  if (!interleave) {
    const unsigned long long hit = __ballot(split >= incl - ns && split < incl);  // at most one lane
    if (!hit) { *beg = 0; *end = 0; *gsid = split; return; }
    src = __ffsll((long long)hit) - 1;
    sidx = split - __shfl(incl - ns, src, 64);
  } else {
    const int total = __shfl(incl, 63, 64);
    if (interleave == 2) {
      // XCD-aware: workgroup i runs on XCD i % 8 (observed placement); give XCD x the x-th CONTIGUOUS eighth of the position-major
      // sequence, so that all offsets of a stretch of rows meet in ONE L2 (the grid is padded to 8 ceil(total / 8) workgroups)
      const int chunk = (total + 7) >> 3;
      split = (split & 7) * chunk + (split >> 3);
    }
    if (split >= total) { *beg = 0; *end = 0; *gsid = 0; return; }
    int rmax = ns;
    for (int o = 32; o > 0; o >>= 1) { const int t = __shfl_xor(rmax, o, 64); rmax = t > rmax ? t : rmax; }
    // C(r) = workgroups in rounds 0 .. r = sum over the non-empty offsets of floor(r ns_k / R) + 1; smallest r with C(r) > split
    auto upto = [&](int r) {
      int c = ns > 0 ? (int)(((long long)r * ns) / rmax) + 1 : 0;
      for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
      return c;
    };
    int a = 0, b = rmax - 1;
    while (a < b) {
      const int m = (a + b) >> 1;
      if (upto(m) > split) b = m; else a = m + 1;
    }
    const int r = a;
    const int before = r > 0 ? upto(r - 1) : 0;
    const int s_r = ns > 0 ? (int)(((long long)r * ns) / rmax) : -1;
    const int s_p = (ns > 0 && r > 0) ? (int)(((long long)(r - 1) * ns) / rmax) : -1;
    unsigned long long fresh = __ballot(ns > 0 && s_r != s_p);
    int want = split - before;   // the want-th (ascending offset) fresh split of round r
    while (want > 0) { fresh &= fresh - 1; --want; }
    src = __ffsll((long long)fresh) - 1;
    sidx = __shfl(s_r, src, 64);
  }
  const int b0 = lo + sidx * pch;
  const int e0 = b0 + pch < hi ? b0 + pch : hi;
  *beg = __shfl(b0, src, 64);
  *end = __shfl(e0, src, 64);
  *gsid = __shfl(incl - ns, src, 64) + sidx;
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
  if (threadIdx.x < 64) {  // first split and split count of offset k: one wave scan over the offsets (K <= 64)
    const int lane = threadIdx.x;
    int acc = 0, mine = 0;
    if (K <= 64) {
      const int nsq = lane < K ? (koff[lane + 1] - koff[lane] + pch - 1) / pch : 0;
      int incl = nsq;
      for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
      }
      acc = __shfl(incl - nsq, k, 64);
      mine = __shfl(nsq, k, 64);
    } else {
      for (int q = 0; q < k; ++q) acc += (koff[q + 1] - koff[q] + pch - 1) / pch;
      mine = (koff[k + 1] - koff[k] + pch - 1) / pch;
    }
    if (lane == 0) { sh[0] = acc; sh[1] = mine; }
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
// Register-only form (16-byte-granular shapes): wave-autonomous, operands straight from HBM/L2 into MFMA layout,
// no LDS, no barrier.
//   gW[k][a][b] = sum over the pairs p of offset k:  fa[ia_p][a] * fb[ib_p][b]
// MFMA 16x16x4 with the PAIR axis as the contraction: lane (n = lane&15, g = lane>>4) holds, for
// pair 4j+g, the 16-byte pieces fa[ia][a0+4n .. +3] and fb[ib][b0+4n .. +3]; component f of the
// A piece and component h of the B piece feed output tile (f,h), whose rows/cols are the
// interleaved channels {a0+4i+f} x {b0+4n+h}. One 64x64 output block = 16 tiles = 16 MFMAs per
// TWO 16-byte loads per lane. A workgroup = 4 waves = 4 output blocks of one (offset, pair chunk)
// split; partial blocks go to the workspace and wgrad_reduce_kernel sums the splits in order.
// Measured and dropped (round 2, tools/wgrad_sweep.py): the same MFMA sequence fed from a double-buffered LDS image of
// 32-pair batches gathered once per workgroup ("wgrad4": half the global loads per MFMA, one barrier per 128 MFMAs,
// two workgroups per CU) -- bit-identical results, 7-14 % SLOWER on the >= 96-channel layers and 2.5x slower on thin
// ones. The PCS_ABLATEW debug builds show where wgrad2 stands: stride 8 256 x 256 1048 us with MFMAs only, 770 us
// with loads only, 1185 us together; three waves per SIMD, each one batch ahead, already hide most of the rest (a third
// batch buffer -- rows two batches ahead -- is +1 % on the 48-wide blocks and -10 % on the 64-wide ones, which drop
// to two waves per SIMD).
// ================================================================================================
struct Wgrad2Args {
  const void *fa;  // rows of ET (float, or bf16 / fp16 halfs converted to fp32 on load)
  const void *fb;
  const int32_t *pairs;
  const int32_t *koff;
  float *partial;  // [nsplit_total][ca][cb]
  int ca, cb, K, a_col, pch, nbg;  // nbg = number of b-groups
  int interleave;                  // launch order of the splits (find_split_wave)
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

// N consecutive channels of one row -> a float vector. Fp32: one 4..16-byte load; halfs: one 2..8-byte load (three
// 2-byte loads for the 48-wide groups, whose 6-byte pieces are only 2-byte aligned), widened in registers.
template <typename V, int N> __device__ __forceinline__ V wload(Fp32, const void *base, int64_t elem) {
  return *reinterpret_cast<const V *>(reinterpret_cast<const float *>(base) + elem);
}
template <typename V, int N, typename HT> __device__ __forceinline__ V wload(HT, const void *base, int64_t elem) {
  const uint16_t *p = reinterpret_cast<const uint16_t *>(base) + elem;
  V v;
  if constexpr (N == 4) {
    const uint2 r = *reinterpret_cast<const uint2 *>(p);
    v.x = h2f(HT{}, (uint16_t)(r.x & 0xFFFFu)); v.y = h2f(HT{}, (uint16_t)(r.x >> 16));
    v.z = h2f(HT{}, (uint16_t)(r.y & 0xFFFFu)); v.w = h2f(HT{}, (uint16_t)(r.y >> 16));
  } else if constexpr (N == 3) {
    v.x = h2f(HT{}, p[0]); v.y = h2f(HT{}, p[1]); v.z = h2f(HT{}, p[2]);
  } else if constexpr (N == 2) {
    const uint32_t r = *reinterpret_cast<const uint32_t *>(p);
    v.x = h2f(HT{}, (uint16_t)(r & 0xFFFFu)); v.y = h2f(HT{}, (uint16_t)(r >> 16));
  } else {
    v = h2f(HT{}, p[0]);
  }
  return v;
}

template <int AW, int BW, typename ET>
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
      bt.a[j] = wload<AV, NA>(ET{}, w.fa, (int64_t)ra * w.ca + acl);
      bt.b[j] = wload<BV, NB>(ET{}, w.fb, (int64_t)rb * w.cb + bcl);
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
#ifndef PCS_ABLATEW
#define PCS_ABLATEW 0  /* debug builds: 1 no row loads inside the loop, 2 no MFMAs */
#endif
#if PCS_ABLATEW == 1
  load_rows(b1s, prn);
#endif
  for (int p0 = beg; p0 < end; p0 += 32) {
    const int p2 = p0 + 32 < end ? p0 + 32 : p0;  // clamped: a redundant batch is masked out in mfma_batch
    const int p3 = p0 + 48 < end ? p0 + 48 : p0;
#if PCS_ABLATEW != 1
    load_rows(b1s, prn);
#endif
    prn = load_pairs(p2);
    __builtin_amdgcn_sched_barrier(0);
#if PCS_ABLATEW != 2
    mfma_batch(b0s, p0);
#else
    acc[0][0][0] += wcomp(b0s.a[0], 0) + wcomp(b0s.a[1], 0) + wcomp(b0s.a[2], 0) + wcomp(b0s.a[3], 0) + wcomp(b0s.b[0], 0) + wcomp(b0s.b[1], 0) + wcomp(b0s.b[2], 0) + wcomp(b0s.b[3], 0);
#endif
    __builtin_amdgcn_sched_barrier(0);
#if PCS_ABLATEW != 1
    load_rows(b0s, prn);
#endif
    prn = load_pairs(p3);
    __builtin_amdgcn_sched_barrier(0);
#if PCS_ABLATEW != 2
    if (p0 + 16 < end) mfma_batch(b1s, p0 + 16);  // wave-uniform
#else
    acc[0][0][1] += wcomp(b1s.a[0], 0) + wcomp(b1s.a[1], 0) + wcomp(b1s.a[2], 0) + wcomp(b1s.a[3], 0) + wcomp(b1s.b[0], 0) + wcomp(b1s.b[1], 0) + wcomp(b1s.b[2], 0) + wcomp(b1s.b[3], 0);
#endif
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

template <typename ET>
__global__ void __launch_bounds__(256, 3) wgrad2_kernel(Wgrad2Args w) {
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  int beg, end, gsid;
  find_split_wave(w.koff, w.K, w.pch, blockIdx.x, lane, &beg, &end, &gsid, w.interleave);  // per wave: no LDS, no barrier
  // the four waves of a workgroup own a 2 x 2 patch of output blocks: each A slice and each B slice they gather is
  // shared by two of them through the vector L1 (a 1 x 4 row shared one A slice and read four B slices: 5 slice
  // streams per workgroup instead of 4, +20 % L2 traffic at 256 columns)
  const int nag = wg_ngroups(w.ca);
  const int nsb = (w.nbg + 1) >> 1;
  const int sa = blockIdx.y / nsb, sb = blockIdx.y - sa * nsb;
  const int ag = 2 * sa + (wid >> 1), bg = 2 * sb + (wid & 1);
  if (beg >= end || ag >= nag || bg >= w.nbg) return;
  const int aw = wg_gwidth(w.ca), bw = wg_gwidth(w.cb);
  float *out = w.partial + (int64_t)gsid * w.ca * w.cb;
  const int a0 = aw * ag, b0 = bw * bg;
#define PCS_WG_CASE(A, B) if (aw == A && bw == B) { wgrad_block<A, B, ET>(w, a0, b0, beg, end, out, lane); return; }
  PCS_WG_CASE(64, 64) PCS_WG_CASE(64, 48) PCS_WG_CASE(64, 32) PCS_WG_CASE(64, 16)
  PCS_WG_CASE(48, 64) PCS_WG_CASE(48, 48) PCS_WG_CASE(48, 32) PCS_WG_CASE(48, 16)
  PCS_WG_CASE(32, 64) PCS_WG_CASE(32, 48) PCS_WG_CASE(32, 32) PCS_WG_CASE(32, 16)
  PCS_WG_CASE(16, 64) PCS_WG_CASE(16, 48) PCS_WG_CASE(16, 32) PCS_WG_CASE(16, 16)
#undef PCS_WG_CASE
}

// ================================================================================================
// wgrad3: the weight gradient on the 16-bit MFMAs (v_mfma_f32_16x16x32_{bf16,f16}, fp32 accumulate).
//   gW[k][a][b] = sum over the pairs p of offset k:  fa[ia_p][a] * fb[ib_p][b]
// The contraction runs over PAIRS, and a 16-bit MFMA wants 8 consecutive contraction indices packed in each lane's
// registers -- 8 pairs of ONE channel -- while a gathered row holds consecutive channels of ONE pair. So the rows of a
// 32-pair batch are staged through LDS row-major (coalesced 16-byte row pieces in, one batch shared by the four waves
// of the workgroup = a 128 x 128 block of the weight gradient instead of four private 64 x 64 gathers) and the operand
// fragments are read TRANSPOSED, eight 2-byte reads per fragment, from a chunk-swizzled image (16-channel chunk c of
// pair p sits at chunk c ^ ((p >> 3) & 3): the four lane groups of a read land in four distinct 8-bank ranges).
//   * half operands (bf16 / fp16, mixed precision): one plane, 16 MFMAs per wave and batch;
//   * fp32 operands: each value is split on the fly into THREE bf16 planes (hi, mid, lo with RNE at every step: exact
//     to 2^-26) and six of the nine plane products are accumulated (hi*hi, hi*mid, mid*hi, mid*mid, hi*lo, lo*hi; the
//     three dropped ones are below 2^-25 relative) -- fp32-grade results (measured error vs fp64 below the fp32 MFMA's
//     own) at 6/16 of the fp32-MFMA cost: gfx950 has no TF32 path, its fp32 MFMA runs at 1/16 of the bf16 rate.
// Split plan, workspace layout and the fixed-order split reduction are those of wgrad2 (deterministic).
// ================================================================================================
constexpr int W3_PB = 32;    // pairs per batch = one MFMA contraction step
constexpr int W3_ROW = 128;  // halfs per staged row and operand (two channel groups of <= 64)

typedef __bf16 w3_bf16x2 __attribute__((ext_vector_type(2)));
typedef float w3_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 w3_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 w3_f16x8 __attribute__((ext_vector_type(8)));

struct Wgrad3Args {
  const void *fa;
  const void *fb;
  const int32_t *pairs;
  const int32_t *koff;
  float *partial;  // [nsplit_total][ca][cb]
  int ca, cb, K, a_col, pch;
  int aw, bw, nag, nbg, nsb;  // group widths, group counts, b super-groups (pairs of groups) per row of super-blocks
  int interleave;             // launch order of the splits (find_split_wave)
  int nsy = 1;                // super-blocks per split (interleave == 2: the launch is one-dimensional)
};

template <typename ET> struct W3Mode;
template <> struct W3Mode<Fp32> { static constexpr int PLANES = 3; using MT = Bf16; };
template <> struct W3Mode<Bf16> { static constexpr int PLANES = 1; using MT = Bf16; };
template <> struct W3Mode<Fp16> { static constexpr int PLANES = 1; using MT = Fp16; };

__device__ __forceinline__ f32x4 w3_mfma(Bf16, const uint4 &a, const uint4 &b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(w3_bf16x8, a), __builtin_bit_cast(w3_bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 w3_mfma(Fp16, const uint4 &a, const uint4 &b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(w3_f16x8, a), __builtin_bit_cast(w3_f16x8, b), c, 0, 0, 0);
}

// (x, y) fp32 -> packed bf16 pairs of the three planes (v_cvt_pk_bf16_f32: round to nearest even)
__device__ __forceinline__ void w3_split(float x, float y, uint32_t &hi, uint32_t &mid, uint32_t &lo) {
  const w3_f32x2 v = {x, y};
  const w3_bf16x2 h = __builtin_convertvector(v, w3_bf16x2);
  const w3_f32x2 r1 = v - __builtin_convertvector(h, w3_f32x2);
  const w3_bf16x2 m = __builtin_convertvector(r1, w3_bf16x2);
  const w3_f32x2 r2 = r1 - __builtin_convertvector(m, w3_f32x2);
  const w3_bf16x2 l = __builtin_convertvector(r2, w3_bf16x2);
  hi = __builtin_bit_cast(uint32_t, h); mid = __builtin_bit_cast(uint32_t, m); lo = __builtin_bit_cast(uint32_t, l);
}

// NT / ROW [r5]: the THIN instance (NT = 64, ROW = 64) -- layers whose whole gradient block is one 64 x 64 tile (32 -> 32, 64 -> 64:
// ten launches per MinkUNet-34 step, 2.5 ms on wgrad2's fp32 MFMAs under autocast) -- is the same kernel as ONE-wave workgroups with
// a 64-half staged row: each wave stages and contracts its own split (no idle waves: of the 4-wave form's 2 x 2 blocks three would
// not exist), 16 KB of LDS per workgroup, four times as many (shorter) splits to keep the CUs full.
template <typename ET, int NA, int NB, int NT = 256, int ROW = W3_ROW>
__global__ void __launch_bounds__(NT, NT == 256 ? (W3Mode<ET>::PLANES == 1 ? 3 : 2) : 4) wgrad3_kernel(Wgrad3Args w) {
  constexpr int PL = W3Mode<ET>::PLANES;
  using MT = typename W3Mode<ET>::MT;
  constexpr bool F32IN = std::is_same<ET, Fp32>::value;
  constexpr int NBUF = PL == 1 ? 2 : 1;  // halfs: two images (16 KB each), one barrier per batch; fp32: one 48 KB image
  constexpr int IMG = PL * 2 * W3_PB * ROW;  // halfs per image: [plane][A | B][pair][ROW]
  __shared__ __attribute__((aligned(16))) uint16_t lds[NBUF * IMG];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int g = lane >> 4, l15 = lane & 15;
  int beg, end, gsid;
  int bx = blockIdx.x, by = blockIdx.y;
  if (w.interleave == 2 && w.nsy > 1) {
    // one-dimensional launch: on every XCD (workgroup id % 8) the super-blocks of one split run back to back, so the rows a split
    // gathers for its first 128 x 128 block serve the others out of that XCD's L2 (256-channel layers: four blocks per split)
    const int slot = bx >> 3;
    by = slot % w.nsy;
    bx = ((slot / w.nsy) << 3) | (bx & 7);
  }
  find_split_wave(w.koff, w.K, w.pch, bx, lane, &beg, &end, &gsid, w.interleave);  // the same answer in every wave
  if (beg >= end) return;  // workgroup-uniform
  const int sa = by / w.nsb, sb = by - sa * w.nsb;  // super-block = 2 x 2 channel groups
  const int a_base = 2 * sa * w.aw, b_base = 2 * sb * w.bw;
  const int a_stage = (w.ca - a_base) < 2 * w.aw ? (w.ca - a_base) : 2 * w.aw;  // channels staged per operand
  const int b_stage = (w.cb - b_base) < 2 * w.bw ? (w.cb - b_base) : 2 * w.bw;
  const int ag = 2 * sa + (wid >> 1), bg = 2 * sb + (wid & 1);
  const bool active = ag < w.nag && bg < w.nbg;  // wave-uniform: this wave owns an output block
  const int aoff = (wid >> 1) * w.aw, boff = (wid & 1) * w.bw;

  f32x4 acc[NA][NB];
#pragma unroll
  for (int i = 0; i < NA; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j) acc[i][j] = (f32x4){0, 0, 0, 0};

  // staging: the image is PAIR-INTERLEAVED -- one 32-bit word holds channel ch of the two pairs (2q, 2q + 1) -- so that a
  // transposed fragment is FOUR ds_read_b32 (8 pairs of one channel) instead of eight 2-byte reads plus packing. A thread
  // therefore moves the same 16-byte channel piece of BOTH rows of a pair couple: halfs are zipped with v_perm_b32, fp32
  // values are converted two at a time (v_cvt_pk_bf16_f32 packs (pair 2q, pair 2q + 1) into exactly that word).
  constexpr int CPP = F32IN ? 4 : 8;                 // channels per 16-byte piece
  constexpr int PPR = ROW / CPP;                     // pieces per staged row: 32 / 16
  constexpr int PIECES = (W3_PB / 2) * PPR / NT;     // (couple, piece) items per thread and operand: 2 / 1
  static_assert(PIECES >= 1 && (W3_PB / 2) * PPR % NT == 0, "staging items must divide over the threads");
  uint32_t *ldw = reinterpret_cast<uint32_t *>(lds);
  constexpr int IMGW = IMG / 2;                      // words per image
  constexpr int OPW = (W3_PB / 2) * ROW;             // words per (plane, operand): 16 couples x ROW channels
  // word position of channel ch of couple q: 16-channel chunk c sits at chunk c ^ ((q >> 2) & 3), so the four lane groups
  // of a fragment read (couples 4 g + j) hit four distinct 16-bank ranges
  auto wpos = [](int q, int ch) { return q * ROW + ((((ch >> 4) ^ (q >> 2)) & (ROW / 16 - 1)) << 4) + (ch & 15); };
  // the dependent chain pair -> row address -> row is cut in two: the pair indices run one batch ahead of the rows
  int2 prs[PIECES][2];
  uint4 ra[PIECES][2], rb[PIECES][2];
  auto fetch_pairs = [&](int p0) {
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
      const int q = (tid + NT * i) / PPR;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        prs[i][h] = make_int2(-1, -1);
        if (p0 + 2 * q + h < end) prs[i][h] = reinterpret_cast<const int2 *>(w.pairs)[p0 + 2 * q + h];
      }
    }
  };
  auto fetch_rows = [&]() {
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
      const int c = ((tid + NT * i) % PPR) * CPP;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const bool pv = prs[i][h].x >= 0;
        const int ia = w.a_col ? prs[i][h].y : prs[i][h].x, ib = w.a_col ? prs[i][h].x : prs[i][h].y;
        ra[i][h] = make_uint4(0u, 0u, 0u, 0u);
        rb[i][h] = make_uint4(0u, 0u, 0u, 0u);
        if (pv && c < a_stage) {
          if (F32IN) ra[i][h] = *reinterpret_cast<const uint4 *>(reinterpret_cast<const float *>(w.fa) + (int64_t)ia * w.ca + a_base + c);
          else ra[i][h] = *reinterpret_cast<const uint4 *>(reinterpret_cast<const uint16_t *>(w.fa) + (int64_t)ia * w.ca + a_base + c);
        }
        if (pv && c < b_stage) {
          if (F32IN) rb[i][h] = *reinterpret_cast<const uint4 *>(reinterpret_cast<const float *>(w.fb) + (int64_t)ib * w.cb + b_base + c);
          else rb[i][h] = *reinterpret_cast<const uint4 *>(reinterpret_cast<const uint16_t *>(w.fb) + (int64_t)ib * w.cb + b_base + c);
        }
      }
    }
  };
  auto stage = [&](int buf) {
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
      const int e = tid + NT * i, q = e / PPR, c = (e % PPR) * CPP;
      const int pos = wpos(q, c);
#pragma unroll
      for (int op = 0; op < 2; ++op) {
        const uint4 v0 = op ? rb[i][0] : ra[i][0], v1 = op ? rb[i][1] : ra[i][1];
        uint32_t *base = ldw + buf * IMGW + op * OPW + pos;
        if (F32IN) {  // 4 channels x 2 pairs -> one 16-byte store per plane
          uint4 hi, mid, lo;
          w3_split(__uint_as_float(v0.x), __uint_as_float(v1.x), hi.x, mid.x, lo.x);
          w3_split(__uint_as_float(v0.y), __uint_as_float(v1.y), hi.y, mid.y, lo.y);
          w3_split(__uint_as_float(v0.z), __uint_as_float(v1.z), hi.z, mid.z, lo.z);
          w3_split(__uint_as_float(v0.w), __uint_as_float(v1.w), hi.w, mid.w, lo.w);
          *reinterpret_cast<uint4 *>(base) = hi;
          *reinterpret_cast<uint4 *>(base + 2 * OPW) = mid;
          *reinterpret_cast<uint4 *>(base + 4 * OPW) = lo;
        } else {      // 8 channels x 2 pairs: zip the halfs of the two rows (even pair in the low half of the word)
          uint4 w0, w1;
          w0.x = __builtin_amdgcn_perm(v1.x, v0.x, 0x05040100u); w0.y = __builtin_amdgcn_perm(v1.x, v0.x, 0x07060302u);
          w0.z = __builtin_amdgcn_perm(v1.y, v0.y, 0x05040100u); w0.w = __builtin_amdgcn_perm(v1.y, v0.y, 0x07060302u);
          w1.x = __builtin_amdgcn_perm(v1.z, v0.z, 0x05040100u); w1.y = __builtin_amdgcn_perm(v1.z, v0.z, 0x07060302u);
          w1.z = __builtin_amdgcn_perm(v1.w, v0.w, 0x05040100u); w1.w = __builtin_amdgcn_perm(v1.w, v0.w, 0x07060302u);
          *reinterpret_cast<uint4 *>(base) = w0;
          *reinterpret_cast<uint4 *>(base + 4) = w1;
        }
      }
    }
  };
  // transposed fragment: lane (n = l15, g) <- channel ch of the staged pairs 8 g .. 8 g + 7 = couples 4 g .. 4 g + 3
  auto frag = [&](int buf, int plane, int op, int ch) {
    const uint32_t *q = ldw + buf * IMGW + (plane * 2 + op) * OPW + wpos(4 * g, ch);
    uint4 f;
    f.x = q[0 * ROW]; f.y = q[1 * ROW]; f.z = q[2 * ROW]; f.w = q[3 * ROW];
    return f;
  };
  auto compute = [&](int buf) {
    uint4 bf[NB][PL];
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int pl = 0; pl < PL; ++pl) bf[j][pl] = frag(buf, pl, 1, boff + 16 * j + l15);
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      uint4 af[PL];
#pragma unroll
      for (int pl = 0; pl < PL; ++pl) af[pl] = frag(buf, pl, 0, aoff + 16 * i + l15);
      // term-major, column-minor: NB independent accumulators between two MFMAs on the same one
      if (PL == 3) {  // smallest products first: lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, then hi*hi
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[i][j] = w3_mfma(MT{}, af[PL - 1], bf[j][0], acc[i][j]);
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[i][j] = w3_mfma(MT{}, af[0], bf[j][PL - 1], acc[i][j]);
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[i][j] = w3_mfma(MT{}, af[PL / 2], bf[j][PL / 2], acc[i][j]);
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[i][j] = w3_mfma(MT{}, af[PL / 2], bf[j][0], acc[i][j]);
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[i][j] = w3_mfma(MT{}, af[0], bf[j][PL / 2], acc[i][j]);
      }
#pragma unroll
      for (int j = 0; j < NB; ++j) acc[i][j] = w3_mfma(MT{}, af[0], bf[j][0], acc[i][j]);
    }
  };

  fetch_pairs(beg);
  fetch_rows();
  fetch_pairs(beg + W3_PB);
  stage(0);
  __syncthreads();
  int cur = 0;
  for (int p0 = beg; p0 < end; p0 += W3_PB) {
    const bool more = p0 + W3_PB < end;  // workgroup-uniform
    if (more) {  // next batch's rows fly during this batch's MFMAs; the pair indices of the one after behind them
      fetch_rows();
      fetch_pairs(p0 + 2 * W3_PB);
    }
    if (active) compute(cur);
    if (NBUF == 2) {
      if (more) stage(cur ^ 1);  // the other image: everybody left it before the previous barrier
      __syncthreads();
      cur ^= 1;
    } else {
      __syncthreads();  // all fragment reads done before the image is overwritten
      if (more) stage(0);
      __syncthreads();
    }
  }
  if (!active) return;
  // tile (i, j), register r: row a0 + 16 i + 4 g + r, column b0 + 16 j + l15
  const int a0 = a_base + aoff, b0 = b_base + boff;
  float *out = w.partial + (int64_t)gsid * w.ca * w.cb;
#pragma unroll
  for (int i = 0; i < NA; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = a0 + 16 * i + 4 * g + r;
      if (row < w.ca) {
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          const int col = b0 + 16 * j + l15;
          if (col < w.cb) out[(int64_t)row * w.cb + col] = acc[i][j][r];
        }
      }
    }
}

template <typename ET>
int launch_wgrad3(const Wgrad3Args &w, int ns, hipStream_t st) {
  const int nsa = (w.nag + 1) / 2;
  const int nsy = nsa * w.nsb;
  const dim3 grid = w.interleave == 2 ? dim3((unsigned)(8 * ((ns + 7) / 8) * nsy), 1u) : dim3((unsigned)ns, (unsigned)nsy);
  const int na = w.aw / 16, nb = w.bw / 16;
#define PCS_W3_CASE(A, B) if (na == A && nb == B) { hipLaunchKernelGGL((wgrad3_kernel<ET, A, B>), grid, dim3(256), 0, st, w); return check_launch("pcs_conv_wgrad(wgrad3)"); }
  PCS_W3_CASE(4, 4) PCS_W3_CASE(4, 3) PCS_W3_CASE(4, 2) PCS_W3_CASE(4, 1)
  PCS_W3_CASE(3, 4) PCS_W3_CASE(3, 3) PCS_W3_CASE(3, 2) PCS_W3_CASE(3, 1)
  PCS_W3_CASE(2, 4) PCS_W3_CASE(2, 3) PCS_W3_CASE(2, 2) PCS_W3_CASE(2, 1)
  PCS_W3_CASE(1, 4) PCS_W3_CASE(1, 3) PCS_W3_CASE(1, 2) PCS_W3_CASE(1, 1)
#undef PCS_W3_CASE
  set_error("pcs_conv_wgrad(wgrad3): unreachable");
  return PCS_EINVAL;
}

// thin half-operand shapes: the whole gradient block is one <= 64 x 64 tile -> one-wave workgroups (wgrad3_kernel<.., 64, 64>)
inline bool wgrad3_thin(int ca, int cb, int dtype) {
  static const int on = getenv("PCS_WGRAD3_THIN") ? atoi(getenv("PCS_WGRAD3_THIN")) : 1;  // 0: wgrad2 as before (A/B)
  return on && dtype != 0 && ca % 16 == 0 && cb % 16 == 0 && ca <= 64 && cb <= 64 && ca >= 32 && cb >= 32;
}

template <typename ET>
int launch_wgrad3_thin(const Wgrad3Args &w, int ns, hipStream_t st) {
  const dim3 grid((unsigned)(w.interleave == 2 ? 8 * ((ns + 7) / 8) : ns), 1);
  const int na = w.aw / 16, nb = w.bw / 16;
#define PCS_W3T_CASE(A, B) if (na == A && nb == B) { hipLaunchKernelGGL((wgrad3_kernel<ET, A, B, 64, 64>), grid, dim3(64), 0, st, w); return check_launch("pcs_conv_wgrad(wgrad3 thin)"); }
  PCS_W3T_CASE(4, 4) PCS_W3T_CASE(4, 3) PCS_W3T_CASE(4, 2) PCS_W3T_CASE(3, 4) PCS_W3T_CASE(3, 3) PCS_W3T_CASE(3, 2)
  PCS_W3T_CASE(2, 4) PCS_W3T_CASE(2, 3) PCS_W3T_CASE(2, 2)
#undef PCS_W3T_CASE
  set_error("pcs_conv_wgrad(wgrad3 thin): unreachable");
  return PCS_EINVAL;
}

int wgrad_plan(const int32_t *koff_host, int K, int ca, int cb, int *pch_out, bool thin = false, bool half = false) {
  // workgroup = 4 output blocks of one split; >= 64 pairs per split
  const int nbq = (wg_ngroups(ca) * wg_ngroups(cb) + 3) / 4;
  int64_t P = koff_host[K] - koff_host[0];
  // measured per shape (tools/conv_microbench.py): ~3072 workgroups when an offset's weight block needs >= 4 of them
  // (256-channel layers), ~1536 otherwise (64..128 channels: +2..16 %; fewer, longer splits = less partial traffic)
  static const int tgt_env = getenv("PCS_WGRAD_TARGET") ? atoi(getenv("PCS_WGRAD_TARGET")) : 0;  // debug: workgroups per launch
  // [r6] 16-bit operands on the 256-channel layers: ~2048 (profiles/round6_wgrad_target_sweep.txt: 256 x 256 0.289 -> 0.248 ms at
  // stride 8, 0.144 -> 0.121 ms at stride 16 -- the kernel is 3x shorter than the fp32 one, the partial sums weigh more)
  int64_t target = (tgt_env > 0 ? tgt_env : (nbq >= 4 ? (half ? 2048 : 3072) : 1536)) / nbq;
  if (thin) target *= 4;   // one-wave workgroups: as many waves in flight as the four-wave form
  if (target < K) target = K;
  int pch = (int)ceil_div(P > 0 ? P : 1, target);
  pch = (int)(ceil_div(pch, 32) * 32);
  if (pch < 64) pch = 64;
  int64_t ns = 0;
  for (int k = 0; k < K; ++k) ns += ceil_div((int64_t)koff_host[k + 1] - koff_host[k], pch);
  *pch_out = pch;
  return (int)ns;
}

}  // namespace

extern "C" size_t pcs_conv_wgrad_ws_bytes(const int32_t *koff_host, int32_t K, int32_t ca,
                                          int32_t cb) {
  if (!koff_host || K <= 0 || ca <= 0 || cb <= 0) return 0;
  int pch;
  int ns = wgrad_plan(koff_host, K, ca, cb, &pch);
  if (wgrad3_thin(ca, cb, 1)) {   // the query does not know the operand type: room for the thin plan's (more, shorter) splits too
    const int ns_thin = wgrad_plan(koff_host, K, ca, cb, &pch, true);
    ns = ns_thin > ns ? ns_thin : ns;
  }
  const int ns_half = wgrad_plan(koff_host, K, ca, cb, &pch, false, true);
  ns = ns_half > ns ? ns_half : ns;
  return (size_t)(ns > 0 ? ns : 1) * ca * cb * sizeof(float);
}

static int g_wgrad_interleave = -1;   // debug / A-B override of PCS_WGRAD_INTERLEAVE (pcs_debug_wgrad_interleave)
extern "C" void pcs_debug_wgrad_interleave(int32_t mode) { g_wgrad_interleave = mode; }

// dtype 0: fp32 operands; 1 / 2: bf16 / fp16 operands (the weight gradient is accumulated and returned in fp32)
static int conv_wgrad_any(const void *fa_v, int32_t ca, const void *fb_v, int32_t cb,
                          const int32_t *pairs, int32_t a_col, const int32_t *koff_dev,
                          const int32_t *koff_host, int32_t K, float *gW, void *ws,
                          size_t ws_bytes, int dtype, void *stream, bool force_split = false) {
  const float *fa = reinterpret_cast<const float *>(fa_v), *fb = reinterpret_cast<const float *>(fb_v);
  if (ca <= 0 || cb <= 0 || K <= 0 || !koff_dev || !koff_host || !gW || (a_col != 0 && a_col != 1) || dtype < 0 || dtype > 2) {
    set_error("pcs_conv_wgrad_f32: bad args");
    return PCS_EINVAL;
  }
  hipStream_t st = as_stream(stream);
  int pch;
  static const int use3_thin = getenv("PCS_WGRAD3") ? atoi(getenv("PCS_WGRAD3")) : 1;
  const bool thin = use3_thin >= 1 && wgrad3_thin(ca, cb, dtype) && (((uintptr_t)fa_v | (uintptr_t)fb_v) & 15) == 0;
  const int ns = wgrad_plan(koff_host, K, ca, cb, &pch, thin, dtype != 0);
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
  if (dtype != 0 && !vec) { set_error("pcs_conv_wgrad_h: half operands need channel counts that are multiples of 4 and 16-byte aligned tensors"); return PCS_EUNSUPPORTED; }
  // 16-bit MFMA path (wgrad3): 16-byte row pieces in both operands. PCS_WGRAD3=0 keeps the fp32-MFMA kernel (A/B, debug).
  // Policy (tools/wgrad_microbench.py, profiles/round2_wgrad3_microbench.md): half operands with >= 4 output blocks
  // (all four waves of a super-block busy) take wgrad3 -- 2.7-3.4x the fp32-MFMA kernel. fp32 operands keep the exact
  // fp32-MFMA kernel (wgrad2) by default: the three-plane split is 1.3-1.4x faster on the >= 96-channel layers and
  // fp32-grade, but the fp32 path of this library stays on fp32 arithmetic unless the caller opts in
  // (pcs_conv_wgrad_f32_bf16x3, or PCS_WGRAD3=2 for every shape). PCS_WGRAD3=0: wgrad2 always (A/B).
  static const int use3 = getenv("PCS_WGRAD3") ? atoi(getenv("PCS_WGRAD3")) : 1;
  // launch order of the splits (find_split_wave): 0 offset-major, 1 position-major, 2 position-major with every XCD on its own
  // contiguous eighth of the sequence. Measured [r6, profiles/round6_wgrad_interleave_ab{,2}.txt], bit-identical results: 1 is
  // neutral (0.94-1.05x); 2 is 1.04-1.31x on the 16-bit kernel (3.05 -> 2.56 ms over the shapes of a step: all offsets of a stretch
  // of rows meet in ONE L2 instead of the Infinity Cache) and neutral / 0.90x on the MFMA-bound fp32 kernel -> 2 for half operands
  static const int interleave_env = getenv("PCS_WGRAD_INTERLEAVE") ? atoi(getenv("PCS_WGRAD_INTERLEAVE")) : -1;
  const int interleave = g_wgrad_interleave >= 0 ? g_wgrad_interleave : (interleave_env >= 0 ? interleave_env : (dtype != 0 ? 2 : 0));
  const int cgran = dtype == 0 ? 4 : 8;
  const bool want3 = dtype == 0 ? (use3 == 2 || force_split) : (use3 >= 1 && (use3 == 2 || wg_ngroups(ca) * wg_ngroups(cb) >= 4));
  if (thin) {
    Wgrad3Args w3;
    w3.fa = fa_v; w3.fb = fb_v; w3.pairs = pairs; w3.koff = koff_dev; w3.partial = reinterpret_cast<float *>(ws);
    w3.ca = ca; w3.cb = cb; w3.K = K; w3.a_col = a_col; w3.pch = pch;
    w3.aw = wg_gwidth(ca); w3.bw = wg_gwidth(cb); w3.nag = 1; w3.nbg = 1; w3.nsb = 1; w3.interleave = interleave;
    int rc3 = dtype == 1 ? launch_wgrad3_thin<Bf16>(w3, ns, st) : launch_wgrad3_thin<Fp16>(w3, ns, st);
    if (rc3) return rc3;
  } else if (vec && want3 && ca % cgran == 0 && cb % cgran == 0) {
    Wgrad3Args w3;
    w3.fa = fa_v; w3.fb = fb_v; w3.pairs = pairs; w3.koff = koff_dev; w3.partial = reinterpret_cast<float *>(ws);
    w3.ca = ca; w3.cb = cb; w3.K = K; w3.a_col = a_col; w3.pch = pch;
    w3.aw = wg_gwidth(ca); w3.bw = wg_gwidth(cb); w3.nag = wg_ngroups(ca); w3.nbg = wg_ngroups(cb); w3.nsb = (w3.nbg + 1) / 2;
    w3.interleave = interleave;
    w3.nsy = ((w3.nag + 1) / 2) * w3.nsb;
    int rc3 = dtype == 0 ? launch_wgrad3<Fp32>(w3, ns, st) : (dtype == 1 ? launch_wgrad3<Bf16>(w3, ns, st) : launch_wgrad3<Fp16>(w3, ns, st));
    if (rc3) return rc3;
  } else if (vec) {
    Wgrad2Args w2;
    w2.fa = fa; w2.fb = fb; w2.pairs = pairs; w2.koff = koff_dev; w2.partial = reinterpret_cast<float *>(ws);
    w2.ca = ca; w2.cb = cb; w2.K = K; w2.a_col = a_col; w2.pch = pch; w2.nbg = wg_ngroups(cb); w2.interleave = interleave;
    const dim3 grid2((unsigned)(interleave == 2 ? 8 * ((ns + 7) / 8) : ns), (unsigned)(((wg_ngroups(ca) + 1) / 2) * ((wg_ngroups(cb) + 1) / 2)));
    if (dtype == 0) hipLaunchKernelGGL(wgrad2_kernel<Fp32>, grid2, dim3(256), 0, st, w2);
    else if (dtype == 1) hipLaunchKernelGGL(wgrad2_kernel<Bf16>, grid2, dim3(256), 0, st, w2);
    else hipLaunchKernelGGL(wgrad2_kernel<Fp16>, grid2, dim3(256), 0, st, w2);
  } else {
    dim3 grid((unsigned)ns, (unsigned)ceil_div(ca, 128), (unsigned)ceil_div(cb, 128));
    hipLaunchKernelGGL(wgrad_kernel<false>, grid, dim3(256), 0, st, w);
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

extern "C" int pcs_conv_wgrad_f32(const float *fa, int32_t ca, const float *fb, int32_t cb,
                                  const int32_t *pairs, int32_t a_col, const int32_t *koff_dev,
                                  const int32_t *koff_host, int32_t K, float *gW, void *ws,
                                  size_t ws_bytes, void *stream) {
  return conv_wgrad_any(fa, ca, fb, cb, pairs, a_col, koff_dev, koff_host, K, gW, ws, ws_bytes, 0, stream);
}

extern "C" int pcs_conv_wgrad_h(const void *fa, int32_t ca, const void *fb, int32_t cb,
                                const int32_t *pairs, int32_t a_col, const int32_t *koff_dev,
                                const int32_t *koff_host, int32_t K, float *gW, void *ws,
                                size_t ws_bytes, int32_t dtype, void *stream) {
  if (dtype != 1 && dtype != 2) { set_error("pcs_conv_wgrad_h: dtype must be 1 (bf16) or 2 (fp16)"); return PCS_EINVAL; }
  return conv_wgrad_any(fa, ca, fb, cb, pairs, a_col, koff_dev, koff_host, K, gW, ws, ws_bytes, dtype, stream);
}

// fp32 operands on the 16-bit MFMAs: every value split into three bf16 planes, six plane products accumulated in fp32
// (wgrad3). Same arguments and fp32-grade results as pcs_conv_wgrad_f32 (needs ca % 4 == 0, cb % 4 == 0, 16-byte
// aligned tensors); 1.3-1.4x the fp32-MFMA kernel on >= 96-channel layers, slower on thin ones. Its own entry: the
// default fp32 path does fp32 arithmetic.
extern "C" int pcs_conv_wgrad_f32_bf16x3(const float *fa, int32_t ca, const float *fb, int32_t cb,
                                         const int32_t *pairs, int32_t a_col, const int32_t *koff_dev,
                                         const int32_t *koff_host, int32_t K, float *gW, void *ws,
                                         size_t ws_bytes, void *stream) {
  if ((ca % 4) || (cb % 4) || (((uintptr_t)fa | (uintptr_t)fb) & 15)) { set_error("pcs_conv_wgrad_f32_bf16x3: needs 16-byte granular rows"); return PCS_EUNSUPPORTED; }
  return conv_wgrad_any(fa, ca, fb, cb, pairs, a_col, koff_dev, koff_host, K, gW, ws, ws_bytes, 0, stream, true);
}
