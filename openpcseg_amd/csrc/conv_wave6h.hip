// Half-precision fused convolution, weight-stationary form (conv_os6h_kernel). Replaces, under `--amp`, the gather / mm / scatter
// dataflow of TS:torchsparse/backend/convolution/convolution_cuda.cu:53-165 like conv_wave5h.hip does; same output-stationary
// tile (fp32 accumulator in LDS, every dst row written once) and the same prepared weights (MFMA fragment order).
//
// What round 6's ablation builds of conv_wave5h.hip measured (profiles/round6_convh_ablation.md): that kernel waits for its
// GATHERED A rows more than for its weight fragments. It issues the loads of step s + 1 before the MFMAs of step s -- 2 KB of A
// per wave in flight, every 32-channel step exposed to a full fabric latency -- and it cannot run the gathers further ahead
// because the weight fragments travel on the same in-order `vmcnt` stream and are needed one step later: a wave that waits for
// a younger weight load has waited for every older A load.
//
// Here the weights are STATIONARY. The tile's work is the slice-major stream of sub-groups (RS row blocks of 16 pairs of one
// offset); it is cut into NW contiguous, equally long ranges, one per wave. A wave holds the NS x NCTT weight fragments of its
// current offset in registers for the whole run of sub-groups it owns in that slice (a tile of T rows costs 27 + NW slab loads,
// what a slab staged once per slice through LDS would cost, without a ring, flags or a loader wave), so inside a run the
// vector-memory stream carries gathered rows only and ALL A loads of sub-group i + 1 (NS x RS x 1 KB) are issued before the
// MFMAs of sub-group i: a whole sub-group of latency budget, 3-4x the bytes in flight. The next offset's weights are loaded in
// one burst right behind the last MFMAs of a run and land while the commit runs.
// The slice table lives in registers (lane k = offset k; v_readlane), not in LDS: no table barrier in front of the first loads
// and no LDS round trips in the per-sub-group bookkeeping.
//
// Commit: ticket-ordered read-add-write into the LDS tile as in conv_wave5h.hip, in the STATIC order (sub-group i of every wave,
// waves ascending): race-free, bit-reproducible (a fixed fp32 addition order per launch shape).
// Epilogue: a thread owns 8 columns: one 16-byte store per row piece (conv_wave5h.hip: 8-byte stores).
//
// Shapes: cin % 32 == 0 with NS x NCTT <= 32 fragments (instances below); everything else stays on conv_wave5h.hip. Padding rows
// of a row block re-read the slice's last pair and land in the sink row (never read).
// Rejected on the way (profiles/round6_convh_ws.md): units of <= RU row blocks dealt round-robin (v1,
// tools/experimental/csrc/conv_wave6h_v1_units.hip.txt); gathering with 4 consecutive lanes per row + a ds_bpermute transpose
// (0 .. -3 %); 64-column tiles for the 256-channel layers (0.8 - 1.08x); one 8-wave workgroup per CU on 384 rows (the 8-wave
// ticket chain: 33 % of the wave time in the ticket wait).
#include "conv_half.h"

using namespace pcs;

namespace pcs {
int launch_conv_wave6h(const ConvArgsH &a, int dtype, hipStream_t st);
bool conv6h_applies(int cin, int cout, int K);
bool conv6h_is_chunked(int cin, int cout);
int conv6h_mode();
}

namespace {

#if PCS_TRACE
// per (block, wave): t_entry, t_first, t_end, issue, mfma (+ operand waits), ticket [high half: operand wait, PCS_TRACE=2], commit,
// t_exit << 8 | sub-groups   (shader clock ticks)
__device__ long long *g_ws_trace;
__device__ int g_ws_trace_blocks;
#define WS_T(...) __VA_ARGS__
#else
#define WS_T(...)
#endif

template <int NCTT, int NW_>
struct Conv6hCfg {
  static constexpr int NW = NW_;
  static constexpr int NT = 64 * NW;
  static constexpr int CT = 16 * NCTT;
  static constexpr int ACS = CT + 4;
  static constexpr int N4 = NCTT / 4;
  static constexpr int N2 = (NCTT % 4) / 2;
  static constexpr int SINK = kConvSinkRows;
  static constexpr int Q8 = CT / 8;        // epilogue: threads per row (8 columns each)
  static constexpr int NRG8 = NT / Q8;     // row groups
  static constexpr size_t lds_bytes(int T) { return (size_t)((T + SINK) * ACS) * 4 + 64; }
};

// Tile epilogue with 16-byte stores: a thread owns 8 consecutive columns and every NRG-th row; `stats` as conv_tile_epilogue
// (conv_common.h): per tile and column sum(x), sum(x^2) of the values AS STORED, taken about the tile's first row, reduced over
// the row groups in a fixed order through the free tile, un-shifted in double. Needs T >= 2 NRG + 1 rows of scratch for stats.
template <typename HT, int CT, int NT>
__device__ __forceinline__ void half_tile_epilogue8(float *acc_l, int ACS, int rows, int n0, int cout, const float *bias, double *stats,
                                                    int tid, uint16_t *drow, int ldd, const uint16_t *arow, float act_slope) {
  constexpr int Q = CT / 8, NRG = NT / Q;
  const int q = tid % Q, rg = tid / Q, c8 = 8 * q;
  const bool on = rg < NRG && n0 + c8 < cout;   // cout % 8 == 0: a piece is inside or outside as a whole
  float b[8], piv[8], s0[8], s1[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) b[j] = piv[j] = s0[j] = s1[j] = 0.f;
  if (on) {
    if (bias) {
      const float4 b0 = *reinterpret_cast<const float4 *>(bias + n0 + c8), b1 = *reinterpret_cast<const float4 *>(bias + n0 + c8 + 4);
      b[0] = b0.x; b[1] = b0.y; b[2] = b0.z; b[3] = b0.w; b[4] = b1.x; b[5] = b1.y; b[6] = b1.z; b[7] = b1.w;
    }
    if (stats) {  // pivot: the tile's first row (+ bias), the same for every row group
      const float4 p0 = *reinterpret_cast<const float4 *>(acc_l + c8), p1 = *reinterpret_cast<const float4 *>(acc_l + c8 + 4);
      piv[0] = p0.x + b[0]; piv[1] = p0.y + b[1]; piv[2] = p0.z + b[2]; piv[3] = p0.w + b[3];
      piv[4] = p1.x + b[4]; piv[5] = p1.y + b[5]; piv[6] = p1.z + b[6]; piv[7] = p1.w + b[7];
    }
    for (int r = rg; r < rows; r += NRG) {
      const float4 v0 = *reinterpret_cast<const float4 *>(acc_l + r * ACS + c8), v1 = *reinterpret_cast<const float4 *>(acc_l + r * ACS + c8 + 4);
      float v[8] = {v0.x + b[0], v0.y + b[1], v0.z + b[2], v0.w + b[3], v1.x + b[4], v1.y + b[5], v1.z + b[6], v1.w + b[7]};
      if (arow) {  // the addend's rows of this tile (kernel argument: uniform)
        const uint4 ad = *reinterpret_cast<const uint4 *>(arow + (int64_t)r * ldd + c8);
        v[0] += h2f(HT{}, (uint16_t)(ad.x & 0xFFFFu)); v[1] += h2f(HT{}, (uint16_t)(ad.x >> 16));
        v[2] += h2f(HT{}, (uint16_t)(ad.y & 0xFFFFu)); v[3] += h2f(HT{}, (uint16_t)(ad.y >> 16));
        v[4] += h2f(HT{}, (uint16_t)(ad.z & 0xFFFFu)); v[5] += h2f(HT{}, (uint16_t)(ad.z >> 16));
        v[6] += h2f(HT{}, (uint16_t)(ad.w & 0xFFFFu)); v[7] += h2f(HT{}, (uint16_t)(ad.w >> 16));
      }
      if (act_slope != 1.f) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = v[j] < 0.f ? v[j] * act_slope : v[j];
      }
      uint16_t h[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) h[j] = f2h(HT{}, v[j]);
      uint4 o;
      o.x = h[0] | ((uint32_t)h[1] << 16); o.y = h[2] | ((uint32_t)h[3] << 16);
      o.z = h[4] | ((uint32_t)h[5] << 16); o.w = h[6] | ((uint32_t)h[7] << 16);
      *reinterpret_cast<uint4 *>(drow + (int64_t)r * ldd + c8) = o;
      if (stats) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float d = h2f(HT{}, h[j]) - piv[j];
          s0[j] += d; s1[j] += d * d;
        }
      }
    }
  }
  if (stats) {
    __syncthreads();
    if (on) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        acc_l[(2 * rg) * ACS + c8 + j] = s0[j];
        acc_l[(2 * rg + 1) * ACS + c8 + j] = s1[j];
        if (rg == 0) acc_l[(2 * NRG) * ACS + c8 + j] = piv[j];
      }
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

// KC > 1: the contraction is cut into KC chunks of NS steps (cin = 32 NS KC): every (offset, chunk) is a slice of its own -- its
// NS x NCTT weight fragments fit the registers where NS KC x NCTT do not (256 / 384 input channels) -- and commits its partial
// sums; the gathered rows of a sub-group are fetched chunk by chunk, each element once.
template <typename HT, int NCTT, int NS, int RS, int NW, int MINW, int KC = 1>
__global__ void __launch_bounds__(64 * NW, MINW) conv_os6h_kernel(ConvArgsH a) {
  using C = Conv6hCfg<NCTT, NW>;
  constexpr int NST = NS * KC;   // steps of the whole contraction (the stride of the prepared weights)
  static_assert(NCTT % 2 == 0, "even number of 16-column tiles");
  const int T = a.tile_rows;
  WS_T(const long long tr_entry = __builtin_readcyclecounter(); long long tr_issue = 0, tr_mfma = 0, tr_ticket = 0, tr_commit = 0;)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float *acc_l = reinterpret_cast<float *>(smem);                        // [T+SINK][ACS]
  int *commit = reinterpret_cast<int *>(acc_l + (T + C::SINK) * C::ACS);
  const unsigned commit_lds = (unsigned)(size_t)(__attribute__((address_space(3))) int *)commit;
  const unsigned acc_lds = (unsigned)(size_t)(__attribute__((address_space(3))) float *)acc_l;

  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, l15 = lane & 15;
  unsigned bid = blockIdx.x;
  if (a.xcd_remap && !a.order) {
    const unsigned nb = gridDim.x, q = nb >> 3, r = nb & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  int64_t slot = bid / a.ncoltiles;
  int ctile = bid % a.ncoltiles;
  if (a.order) {
    const unsigned xcd = bid & 7, idx = bid >> 3;
    slot = (int64_t)(idx / a.ncoltiles) * 8 + xcd;
    ctile = idx % a.ncoltiles;
    if (slot >= a.ntiles) return;
  }
  const int64_t tile = a.order ? (int64_t)a.order[slot] : slot;
  const int n0 = ctile * C::CT;
  const int64_t row0 = tile * T;
  const int64_t nt1 = a.ntiles + 1;

  // ---- the slice table, in every wave's registers: lane k = offset k ------------------------------------------------------
  int s0_v = 0, m_v = 0;
  if (lane < a.K) {
    s0_v = a.seg[(int64_t)lane * nt1 + tile];
    m_v = a.seg[(int64_t)lane * nt1 + tile + 1] - s0_v;
  }
  {  // zero the tile while the segment loads are in flight
    float4 *z = reinterpret_cast<float4 *>(acc_l);
    const int n4 = (T + C::SINK) * (C::ACS / 4);
    for (int i = tid; i < n4; i += C::NT) z[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tid == 0) *commit = 0;
  }
  const int nsg_v = (((m_v + 15) >> 4) + RS - 1) / RS;   // sub-groups of slice k (per contraction chunk)
  int incl_v = KC * nsg_v;                               // inclusive prefix over the offsets
#pragma unroll
  for (int o = 1; o < 32; o <<= 1) {   // K <= 32
    const int t = __shfl_up(incl_v, o, 64);
    if (lane >= o) incl_v += t;
  }
  const int total = __builtin_amdgcn_readlane(incl_v, 31);                 // sub-groups of the tile
  const int cq = total / NW, cr = total % NW;
  const int g_first = wid * cq + (wid < cr ? wid : cr);                    // this wave's range of the stream
  const int n_sub = cq + (wid < cr ? 1 : 0);

  const int gt0 = ctile * NCTT;
  int btile[NCTT];
#pragma unroll
  for (int t = 0; t < NCTT; ++t) btile[t] = (gt0 + t < a.nt16) ? t : 0;

  // position in the stream: sub-group G = sub-group j of slice (= offset) e; m pairs from pair s0. All wave-uniform (SGPRs).
  struct Pos { int G, e, j, nsg, m, s0, c; };   // c: contraction chunk
  auto locate = [&](Pos &p, int G) {
    const unsigned long long later = __ballot(incl_v > G);   // the first offset whose inclusive prefix passes G holds it
    const int e = (int)__builtin_ctzll(later | (1ULL << 63));
    p.G = G; p.e = e;
    p.nsg = __builtin_amdgcn_readlane(nsg_v, e);
    p.j = G - (__builtin_amdgcn_readlane(incl_v, e) - KC * p.nsg);
    p.c = 0;
    if (KC > 1) {
      while (p.j >= p.nsg) { p.j -= p.nsg; ++p.c; }   // chunk-major inside the slice: all sub-groups of chunk 0, then chunk 1 ...
    }
    p.m = __builtin_amdgcn_readlane(m_v, e);
    p.s0 = __builtin_amdgcn_readlane(s0_v, e);
  };
  auto advance = [&](Pos &p) {  // past the tile's last sub-group the position stays (its loads are repeated, harmless)
    if (p.G + 1 >= total) return;
    if (p.j + 1 < p.nsg) { ++p.G; ++p.j; return; }
    if (KC > 1 && p.c + 1 < KC) { ++p.G; p.j = 0; ++p.c; return; }
    locate(p, p.G + 1);
  };
  struct Ctx {
    const char *srow[RS];
    int dloc[RS];
    int nr;
  };
  auto pair_index = [&](const Pos &p, int *pidx) {
#pragma unroll
    for (int r = 0; r < RS; ++r) {
      const int rk = (p.j * RS + r) * 16 + l15;
      pidx[r] = p.s0 + (rk < p.m ? rk : p.m - 1);
    }
  };
  auto make_ctx = [&](Ctx &cx, const Pos &p, const int2 *pr) {
#pragma unroll
    for (int r = 0; r < RS; ++r) {
      const int rk = (p.j * RS + r) * 16 + l15;
      cx.srow[r] = a.src + ((int64_t)(a.src_col ? pr[r].y : pr[r].x) * a.cin + 32 * NS * p.c + 8 * g) * 2;
      cx.dloc[r] = rk < p.m ? (int)((a.src_col ? pr[r].x : pr[r].y) - row0) : T;
    }
    const int left = ((p.m + 15) >> 4) - p.j * RS;
    cx.nr = left < RS ? left : RS;
  };
  auto wk_of = [&](const Pos &p) { return a.Wp + (((int64_t)p.e * a.nt16 + gt0) * NST + NS * p.c) * 1024 + lane * 16; };
  struct AFrag { uint4 v[NS][RS]; };
  auto load_a = [&](AFrag &f, const Ctx &cx) {
#pragma unroll
    for (int r = 0; r < RS; ++r) {
      if (r < cx.nr) {  // wave-uniform: a row block beyond the slice is not gathered (its results go to the sink row)
#pragma unroll
        for (int s = 0; s < NS; ++s) f.v[s][r] = *reinterpret_cast<const uint4 *>(cx.srow[r] + s * 64);
      }
    }
  };

  uint4 B[NS][NCTT];  // the stationary weight fragments of the current offset
  auto load_b = [&](const char *wk) {
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int t = 0; t < NCTT; ++t) B[s][t] = *reinterpret_cast<const uint4 *>(wk + ((size_t)btile[t] * NST + s) * 1024);
  };
  AFrag A0, A1;
#pragma unroll
  for (int s = 0; s < NS; ++s)
#pragma unroll
    for (int r = 0; r < RS; ++r) A0.v[s][r] = A1.v[s][r] = make_uint4(0u, 0u, 0u, 0u);

  Pos pc, pn;
  pc.G = 0; pc.e = 0; pc.j = 0; pc.nsg = 1; pc.m = 1; pc.s0 = 0; pc.c = 0;
  pn = pc;
  Ctx cur;
  int2 prn[RS];
#pragma unroll
  for (int r = 0; r < RS; ++r) { prn[r] = make_int2(0, 0); cur.srow[r] = a.src; cur.dloc[r] = T; }
  cur.nr = 0;
  if (n_sub > 0) {
    locate(pc, g_first);
    int pidx[RS];
    pair_index(pc, pidx);
    int2 pr[RS];
#pragma unroll
    for (int r = 0; r < RS; ++r) pr[r] = reinterpret_cast<const int2 *>(a.pairs)[pidx[r]];
    load_b(wk_of(pc));
    pn = pc;
    advance(pn);
    pair_index(pn, pidx);
#pragma unroll
    for (int r = 0; r < RS; ++r) prn[r] = reinterpret_cast<const int2 *>(a.pairs)[pidx[r]];
    make_ctx(cur, pc, pr);
    load_a(A0, cur);
  } else {
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int t = 0; t < NCTT; ++t) B[s][t] = make_uint4(0u, 0u, 0u, 0u);
  }
  __syncthreads();   // the tile is zero

  // one sub-group: Ac holds its gathered rows (in flight), An receives the next one's
  auto run_sub = [&](const int i, AFrag &Ac, AFrag &An) {
    WS_T(const long long tr_a = __builtin_readcyclecounter();)
    const int ticket = i * NW + wid;   // static order: sub-group i of every wave, waves ascending
    // the pair rows of sub-group i + 2 first (8 bytes per lane; they must be OLDER than the gathers below, or forming the
    // gather addresses of the next sub-group would wait for this one's rows), then the gathers of sub-group i + 1
    Pos pp = pn;
    advance(pp);
    int pidx_p[RS];
    pair_index(pp, pidx_p);
    int2 prp[RS];
#pragma unroll
    for (int r = 0; r < RS; ++r) prp[r] = reinterpret_cast<const int2 *>(a.pairs)[pidx_p[r]];
    Ctx nxt;
    make_ctx(nxt, pn, prn);
    load_a(An, nxt);
    const bool reload = pn.e != pc.e || pn.c != pc.c;  // the next sub-group belongs to another offset / chunk (wave-uniform)

    f32x4 acc[RS][NCTT];
#pragma unroll
    for (int r = 0; r < RS; ++r)
#pragma unroll
      for (int t = 0; t < NCTT; ++t) acc[r][t] = (f32x4){0, 0, 0, 0};
    WS_T(asm volatile("" ::: "memory"); const long long tr_b = __builtin_readcyclecounter();)
#if PCS_TRACE == 2
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // operand wait measured on its own (also waits for the next sub-group's rows)
    const long long tr_b2 = __builtin_readcyclecounter();
#endif
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int r = 0; r < RS; ++r)
#pragma unroll
        for (int t = 0; t < NCTT; ++t) acc[r][t] = mfma_h(HT{}, Ac.v[s][r], B[s][t], acc[r][t]);
    if (reload) load_b(wk_of(pn));   // one burst behind the run's last MFMAs; lands while the commit runs

    // ---- ticket-ordered commit (as conv_wave5h.hip: addresses before the wait, three fenced phases) ----
    unsigned dq[RS][4], dp[RS][4];
#pragma unroll
    for (int r = 0; r < RS; ++r)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int dl = __shfl(cur.dloc[r], 4 * g + j, 64);
        dq[r][j] = acc_lds + 4u * (unsigned)(dl * C::ACS) + 16u * l15;
        dp[r][j] = acc_lds + 4u * (unsigned)(dl * C::ACS) + 256u * C::N4 + 8u * l15;
        asm volatile("" : "+v"(dq[r][j]), "+v"(dp[r][j]));
      }
    const int nr = __builtin_amdgcn_readfirstlane(cur.nr);
#if PCS_TRACE
#pragma unroll
    for (int r = 0; r < RS; ++r)
#pragma unroll
      for (int t = 0; t < NCTT; ++t) asm volatile("" : "+v"(acc[r][t]));   // the MFMAs have retired
    const long long tr_c = __builtin_readcyclecounter();
#endif
    if (lane == 0) {
      while (__hip_atomic_load(commit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != ticket) __builtin_amdgcn_s_sleep(1);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    WS_T(const long long tr_d = __builtin_readcyclecounter();)
    __builtin_amdgcn_s_setprio(3);
    {
      typedef float v2f __attribute__((ext_vector_type(2)));
      typedef __attribute__((address_space(3))) const f32x4 lds_cf4;
      typedef __attribute__((address_space(3))) const v2f lds_cf2;
      typedef __attribute__((address_space(3))) f32x4 lds_f4;
      typedef __attribute__((address_space(3))) v2f lds_f2;
      // all row blocks of the sub-group in one round while the registers allow it (<= 96 columns), else one round per block
      constexpr int RB = 1;   // (two row blocks per round as in conv_wave5h.hip cost 20+ registers this kernel does not have)
#pragma unroll
      for (int r0 = 0; r0 < RS; r0 += RB) {
        f32x4 v4[RB][4][C::N4 > 0 ? C::N4 : 1];
        v2f v2[RB][4];
#pragma unroll
        for (int rr = 0; rr < RB; ++rr)
          if (r0 + rr < nr) {  // wave-uniform
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
              for (int q = 0; q < C::N4; ++q) v4[rr][j][q] = *(lds_cf4 *)(size_t)(dq[r0 + rr][j] + 256u * q);
              if (C::N2) v2[rr][j] = *(lds_cf2 *)(size_t)dp[r0 + rr][j];
            }
          }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int rr = 0; rr < RB; ++rr)
          if (r0 + rr < nr) {
            const int r = r0 + rr;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
              for (int q = 0; q < C::N4; ++q) {
                v4[rr][j][q].x += acc[r][4 * q + 0][j]; v4[rr][j][q].y += acc[r][4 * q + 1][j];
                v4[rr][j][q].z += acc[r][4 * q + 2][j]; v4[rr][j][q].w += acc[r][4 * q + 3][j];
              }
              if (C::N2) { v2[rr][j].x += acc[r][4 * C::N4 + 0][j]; v2[rr][j].y += acc[r][4 * C::N4 + 1][j]; }
            }
          }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int rr = 0; rr < RB; ++rr)
          if (r0 + rr < nr) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
              for (int q = 0; q < C::N4; ++q) *(lds_f4 *)(size_t)(dq[r0 + rr][j] + 256u * q) = v4[rr][j][q];
              if (C::N2) *(lds_f2 *)(size_t)dp[r0 + rr][j] = v2[rr][j];
            }
          }
        asm volatile("" ::: "memory");
      }
    }
    // the ticket store stays behind the tile writes (the LDS keeps one wave's instructions in order, PCS_COMMIT_NOWAIT)
    if (lane == 0) asm volatile("ds_write_b32 %0, %1" ::"v"(commit_lds), "v"(ticket + 1) : "memory");
    __builtin_amdgcn_s_setprio(0);
#if PCS_TRACE
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const long long tr_e = __builtin_readcyclecounter();
    tr_issue += tr_b - tr_a; tr_mfma += tr_c - tr_b; tr_ticket += tr_d - tr_c; tr_commit += tr_e - tr_d;
#if PCS_TRACE == 2
    tr_mfma -= tr_b2 - tr_b; tr_ticket += (tr_b2 - tr_b) << 32;   // the operand wait rides in the high half of [5]
#endif
#endif
    cur = nxt;
    pc = pn;
    pn = pp;
#pragma unroll
    for (int r = 0; r < RS; ++r) prn[r] = prp[r];
  };
  WS_T(const long long tr_first = __builtin_readcyclecounter();)
  {
    int i = 0;
    for (; i + 1 < n_sub; i += 2) {
      run_sub(i, A0, A1);
      run_sub(i + 1, A1, A0);
    }
    if (i < n_sub) run_sub(i, A0, A1);
  }
  WS_T(const long long tr_end = __builtin_readcyclecounter();)
  __syncthreads();
  const int rows = (int)((a.n_dst - row0) < (int64_t)T ? (a.n_dst - row0) : (int64_t)T);
  uint16_t *drow = a.dst + row0 * a.cout + n0;
  const int ldd = a.cout;
  double *stats = a.stats ? a.stats + tile * 2 * a.cout : nullptr;
  const GStat gstat{a.gs_x, a.gs_mask, a.gs_stat, GsType<HT>::value};
  if (!a.gs_x && (a.cout & 7) == 0 && (((uintptr_t)a.dst | (uintptr_t)a.addend) & 15) == 0 && (!stats || T + C::SINK >= 2 * C::NRG8 + 1)) {
    half_tile_epilogue8<HT, C::CT, C::NT>(acc_l, C::ACS, rows, n0, a.cout, a.bias, stats, tid, drow, ldd,
                                          a.addend ? a.addend + row0 * a.cout + n0 : nullptr, a.act_slope);
  } else {
    conv_tile_epilogue<C::CT, C::NT>(acc_l, C::ACS, rows, n0, a.cout, a.bias, stats, tid, [&](int r, int cq4, const float4 &v0) {
      float4 v = v0;
      if (a.addend) {
        const uint2 ad = *reinterpret_cast<const uint2 *>(a.addend + (row0 + r) * (int64_t)ldd + n0 + cq4);
        v.x += h2f(HT{}, (uint16_t)(ad.x & 0xFFFFu)); v.y += h2f(HT{}, (uint16_t)(ad.x >> 16));
        v.z += h2f(HT{}, (uint16_t)(ad.y & 0xFFFFu)); v.w += h2f(HT{}, (uint16_t)(ad.y >> 16));
      }
      if (a.act_slope != 1.f) {
        v.x = v.x < 0.f ? v.x * a.act_slope : v.x; v.y = v.y < 0.f ? v.y * a.act_slope : v.y;
        v.z = v.z < 0.f ? v.z * a.act_slope : v.z; v.w = v.w < 0.f ? v.w * a.act_slope : v.w;
      }
      const uint16_t hx = f2h(HT{}, v.x), hy = f2h(HT{}, v.y), hz = f2h(HT{}, v.z), hw = f2h(HT{}, v.w);
      uint2 o;
      o.x = hx | ((uint32_t)hy << 16);
      o.y = hz | ((uint32_t)hw << 16);
      *reinterpret_cast<uint2 *>(drow + (int64_t)r * ldd + cq4) = o;
      return make_float4(h2f(HT{}, hx), h2f(HT{}, hy), h2f(HT{}, hz), h2f(HT{}, hw));
    }, a.gs_x ? &gstat : nullptr, row0);
  }
#if PCS_TRACE
  if (lane == 0 && g_ws_trace && (int)blockIdx.x < g_ws_trace_blocks) {
    long long *t = g_ws_trace + ((int64_t)blockIdx.x * 8 + wid) * 8;
    t[0] = tr_entry; t[1] = tr_first; t[2] = tr_end; t[3] = tr_issue; t[4] = tr_mfma; t[5] = tr_ticket; t[6] = tr_commit;
    t[7] = (__builtin_readcyclecounter() << 8) | (n_sub & 255);
  }
#endif
}

template <typename HT, int NCTT, int NS, int RS, int NW, int KC = 1>
int launch6h(const ConvArgsH &a, hipStream_t st) {
  using C = Conv6hCfg<NCTT, NW>;
  const int64_t nblocks = a.order ? ceil_div(a.ntiles, 8) * 8 * a.ncoltiles : a.ntiles * a.ncoltiles;
  if (nblocks <= 0) return PCS_OK;
  if (nblocks > 0x7FFFFFFF) { set_error("pcs_conv_h: grid too large"); return PCS_EUNSUPPORTED; }
  auto kern = conv_os6h_kernel<HT, NCTT, NS, RS, NW, 2, KC>;
  const size_t lds = C::lds_bytes(a.tile_rows);
  if (lds > kMaxDynLds) { set_error("pcs_conv_h: tile_rows too large for this column tile"); return PCS_EUNSUPPORTED; }
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxDynLds);
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)nblocks), dim3(C::NT), lds, st, a);
  return check_launch("pcs_conv_gather_gemm_h(wave6h)");
}

// (16-column tiles per column tile, 32-channel steps) the stationary-weight kernel is instantiated for, and the row blocks per
// sub-group of each: NS x NCTT weight fragments of 4 registers stay resident next to RS x NCTT accumulators and two sets of
// NS x RS gathered pieces
struct Ws6Shape { int nctt, ns, rs; };
constexpr Ws6Shape kWs6Shapes[] = {{6, 3, 2}, {6, 4, 1}, {8, 3, 1}, {8, 4, 1}, {4, 2, 2}, {8, 2, 2}, {4, 4, 2}, {6, 2, 2}, {2, 2, 2},
                                   {2, 1, 2}, {4, 1, 2}, {6, 1, 2}, {8, 1, 2}};
inline int conv6h_rs(int nctt, int ns) {
  for (const auto &e : kWs6Shapes)
    if (e.nctt == nctt && e.ns == ns) return e.rs;
  if (nctt == 8 && (ns == 8 || ns == 12 || ns == 6)) return 1;   // chunked contractions (KC = 2 / 3 / 2)
  if (nctt == 6 && ns == 6) return 2;
  if (nctt == 6 && ns == 8) return 1;
  return 0;
}
inline bool conv6h_chunked(int nctt, int ns) { return ns > 4; }
int g_ws_mode = -1, g_ws_rs = 0;  // debug / A-B overrides (pcs_debug_convh_ws); -1 / 0 = environment / default

template <typename HT, int NW>
int dispatch6h(const ConvArgsH &a, int nctt, int rs, hipStream_t st) {
#define PCS_C6H(N, S, R) \
  if (nctt == N && a.ns == S && rs == R) return launch6h<HT, N, S, R, NW>(a, st);
  PCS_C6H(6, 3, 2) PCS_C6H(6, 4, 2) PCS_C6H(8, 3, 2) PCS_C6H(8, 4, 1) PCS_C6H(4, 2, 2) PCS_C6H(8, 2, 2) PCS_C6H(4, 4, 2)
  PCS_C6H(6, 2, 2) PCS_C6H(2, 2, 2) PCS_C6H(2, 1, 2) PCS_C6H(4, 1, 2) PCS_C6H(6, 1, 2) PCS_C6H(8, 1, 2)
  PCS_C6H(6, 4, 1) PCS_C6H(8, 3, 1)
#undef PCS_C6H
  // chunked contractions: 256 / 384 / 192 input channels on 128-column tiles, 192 / 256 on 96-column tiles
  if (nctt == 8 && a.ns == 8) return launch6h<HT, 8, 4, 1, NW, 2>(a, st);
  if (nctt == 8 && a.ns == 12) return launch6h<HT, 8, 4, 1, NW, 3>(a, st);
  if (nctt == 8 && a.ns == 6) return launch6h<HT, 8, 3, 1, NW, 2>(a, st);
  if (nctt == 6 && a.ns == 6) return launch6h<HT, 6, 3, 2, NW, 2>(a, st);
  if (nctt == 6 && a.ns == 8) return launch6h<HT, 6, 4, 1, NW, 2>(a, st);
  set_error("pcs_conv_gather_gemm_h(wave6h): no instance for this shape");
  return PCS_EUNSUPPORTED;
}

}  // namespace

namespace pcs {

bool conv6h_applies(int cin, int cout, int K) {
  if (!convh_applies(cin, cout, K) || cin % 32) return false;
  const int nctt = conv_nctt(cout), ns = cin / 32;
  if (conv6h_rs(nctt, ns) == 0) return false;
  if (conv6h_mode() >= 2 || conv6h_chunked(nctt, ns)) return true;
  // measured on the 12-frame bench maps (profiles/round6_convh_ws.md): 1.2-1.3x on the 64 ... 128-channel layers; the thin
  // shapes (<= 2 weight fragments per step and tile, or one step) stay on conv_wave5h.hip
  // chunked contractions (192 ... 384 input channels): 1.06-1.26x on 4-wave workgroups (tiles <= 160 rows, two per CU), 0.6-0.8x on
  // the 8-wave ones -- the entry point (conv_wave5h.hip) sends only tiles of <= 160 rows here, the picker (conv.hip) asks for 144
  // rows where that pays (everything but 256 -> 256 on the big stride-8 level: 0.97x)
  if (K <= 8) return ns * nctt >= 16 && ns >= 3;   // k = 2 strided / transposed maps: one pair per row and offset
  return ns * nctt >= 8 && ns >= 2 && !(nctt == 8 && ns == 2);
}

int launch_conv_wave6h(const ConvArgsH &a0, int dtype, hipStream_t st) {
  ConvArgsH a = a0;
  const int nctt = conv_nctt(a.cout);
  a.ncoltiles = (int)ceil_div(a.cout, 16 * nctt);
  int rs = conv6h_rs(nctt, a.ns);
  if (g_ws_rs == 1 && ((nctt == 6 && a.ns == 4) || (nctt == 8 && a.ns == 3))) rs = 2;   // A/B: two row blocks (spills a few registers)
  const bool nw8 = 2 * conv5_lds_est(a.tile_rows, nctt) > 160 * 1024;
  if (dtype == 1) return nw8 ? dispatch6h<Bf16, 8>(a, nctt, rs, st) : dispatch6h<Bf16, 4>(a, nctt, rs, st);
  return nw8 ? dispatch6h<Fp16, 8>(a, nctt, rs, st) : dispatch6h<Fp16, 4>(a, nctt, rs, st);
}

// 0: conv_wave5h.hip for everything; 1 (default): the shapes of conv6h_applies() that measured faster (policy below);
// 2: every shape conv6h_applies() serves (A/B)
bool conv6h_is_chunked(int cin, int cout) { return cin % 32 == 0 && conv6h_chunked(conv_nctt(cout), cin / 32); }

int conv6h_mode() {
  static const int ws = getenv("PCS_CONVH_WS") ? atoi(getenv("PCS_CONVH_WS")) : 1;
  return g_ws_mode >= 0 ? g_ws_mode : ws;
}

}  // namespace pcs

// debug / A-B (tools/convh_ws_ab.py): mode -1 environment, 0 off, 1 on; rs = 1: two-row-block sub-groups for the two shapes that
// have both instances (128 -> 96, 96 -> 128)
extern "C" void pcs_debug_convh_ws(int32_t mode, int32_t ru, int32_t rs) {
  (void)ru;
  g_ws_mode = mode; g_ws_rs = rs & 1;
}

#if PCS_TRACE
// debug builds: per-wave phase timers of the next launches go to `buf` ([blocks][8 waves][8] int64, device memory; NULL = off)
extern "C" int pcs_debug_ws_trace(long long *buf, int32_t blocks) {
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_ws_trace), &buf, sizeof(buf)) != hipSuccess) return PCS_ELAUNCH;
  return hipMemcpyToSymbol(HIP_SYMBOL(g_ws_trace_blocks), &blocks, sizeof(blocks)) == hipSuccess ? PCS_OK : PCS_ELAUNCH;
}
#endif
