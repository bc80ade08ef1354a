// Wave-autonomous fused convolution with row-block groups ("wave5"): the kernel behind every >= 64-channel layer.
#include "conv_common.h"

using namespace pcs;

namespace {

#if PCS_TRACE
__device__ long long *g_conv_trace;   // [block][wave][8]: t_entry, t_start, t_end, loop, ticket, commit, groups, t_exit
constexpr int kTraceBlocks = 8192;
#endif

// ================================================================================================
// wave5 = the wave4 design (conv_wave4.hip: register-direct operands in MFMA layout, interleaved column tiles,
// sched_group_barrier software pipeline, cross-group prefetch, ticket-ordered LDS commit) + row-block GROUPS: one wave
// applies each W operand block to a group of up to R consecutive row blocks of the same offset (R accumulator sets),
// so the W stream per compact row drops R-fold where an offset has >= R row blocks in the tile -- with one row block
// per step every 16 rows stream their own copy of W[k] (Cin x CT fp32, 37-131 KB) out of L2. The tile height is a
// run-time parameter chosen per layer (pcs_conv_pick_tile_rows).
// TAIL = false: cin % 32 == 0 (an even number of whole 16-channel blocks: the straight-line two-stage pipeline).
// TAIL = true: any cin % 4 == 0 (56, 112, 168, 336 ... of RPVNet cr 1.75, 48 of cr 0.5): the last block may hold
// 4 / 8 / 12 channels -- its out-of-range lane groups read a clamped (in-row) address and contribute exact zeros --
// and an odd number of blocks ends the group one pipeline stage early (the next group's first block lands in the
// other register set and is moved over).
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
  static constexpr int SINK = kConvSinkRows;  // atomic commit: padding rows of lane group g accumulate into sink row T + g
  static constexpr size_t lds_bytes(int T) { return (size_t)((T + SINK) * ACS) * 4 + 5 * 33 * 4 + 16; }
};

template <int NCTT, int NW, int MINW, int R, bool TAIL>
__global__ void __launch_bounds__(64 * NW, MINW) conv_os5_kernel(ConvArgs a) {
  using C = Conv5Cfg<NCTT, NW, R>;
  const int T = a.tile_rows;  // any multiple of 16: the host picks it per layer (pcs_conv_pick_tile_rows)
  PCS_T(const long long tr_entry = wall_clock64(); long long tr_loop = 0, tr_ticket = 0, tr_commit = 0; int tr_groups = 0;)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float *acc_l = reinterpret_cast<float *>(smem);            // [T+SINK][ACS], rows >= T = sink for padding rows
  int *kl_k = reinterpret_cast<int *>(acc_l + (T + C::SINK) * C::ACS);  // [32] offset id
  int *kl_s = kl_k + 32;                                     // [32] first pair
  int *kl_m = kl_s + 32;                                     // [32] #pairs
  int *kl_g = kl_m + 32;                                     // [33] first FULL group (prefix over the offsets)
  int *kl_h = kl_g + 33;                                     // [33] first partial group (prefix)
  int *commit = kl_h + 33;
  const unsigned commit_lds = (unsigned)(size_t)(__attribute__((address_space(3))) int *)commit;  // LDS byte address
  const unsigned acc_lds = (unsigned)(size_t)(__attribute__((address_space(3))) float *)acc_l;
  __shared__ int nk_s;

  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);  // a scalar: wave-level loops and branches stay uniform
  const int g = lane >> 4, l15 = lane & 15;
  unsigned bid = blockIdx.x;
  if (a.xcd_remap && !a.order) {  // row order: one contiguous tile range per XCD; heaviest-first order: dealt round-robin
    const unsigned nb = gridDim.x, q = nb >> 3, r = nb & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  int64_t slot = bid / a.ncoltiles;
  int ctile = bid % a.ncoltiles;
  if (a.xcd_remap == 2) {  // the column tiles of one row tile on ONE XCD, back to back (they gather the same A rows)
    const unsigned xcd = bid & 7, idx = bid >> 3;
    slot = (int64_t)(idx / a.ncoltiles) * 8 + xcd;
    ctile = idx % a.ncoltiles;
    if (slot >= a.ntiles) return;  // the grid is padded to 8 * ncoltiles
  }
  const int64_t tile = a.order ? (int64_t)a.order[slot] : slot;
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
  {  // zero the tile: (T + SINK) * ACS floats, a multiple of four
    float4 *z = reinterpret_cast<float4 *>(acc_l);
    const int n4 = (T + C::SINK) * (C::ACS / 4);
    for (int i = tid; i < n4; i += C::NT) z[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  __syncthreads();
  const int nk = __builtin_amdgcn_readfirstlane(nk_s);  // scalars: the group loop and its branches are wave-uniform
  // group order = commit order: all full groups (R row blocks, equal duration) in ascending offset order, then the
  // partial groups. Waves take groups round-robin and commit in order, so neighbours of equal length never wait
  // for each other (with offset-major numbering a short group queued behind a long one idled its wave: 9-12 % of
  // the wave time in the ticket wait, tools/conv_trace.py). The order depends on the map only: deterministic.
  const int total_full = nk > 0 ? __builtin_amdgcn_readfirstlane(kl_g[nk]) : 0;
  const int total_grp = nk > 0 ? total_full + __builtin_amdgcn_readfirstlane(kl_h[nk]) : 0;

  const int cin4 = a.cin - 4;
  const int nb16 = TAIL ? (a.cin + 15) >> 4 : a.cin >> 4;  // 16-channel contraction blocks
  const int cinp = nb16 * 16;
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
    int ca = c0 + 4 * g;  // !TAIL: always inside the row
    if (TAIL) ca = ca < cin4 ? ca : cin4;  // a lane group beyond cin re-reads the row's last piece; masked in mfma_frag
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
  // One group = NRC row blocks of one offset: R for the full groups, 1 for the partial ones (R = 2). A lambda per NRC so
  // that each of the two group loops below has ONE straight-line body: with both bodies behind a branch inside one loop the
  // register allocator resolved the loop-carried operand registers (the next group's prefetched first block) with ~30
  // copies at the join -- and the s_waitcnt vmcnt(0) those copies need sat in front of every ticket wait.
  static_assert(R == 2, "partial groups are handled as single row blocks");
  auto run_group = [&](const int grp, auto nrc_tag) {
    constexpr int NRC = decltype(nrc_tag)::value;
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
    constexpr int nr = NRC;
    // MFMAs of one 16-channel block for the first NR row blocks of the group (NR is wave-uniform)
    // last_tag: the block is the layer's last one (only there can lane groups lie beyond cin)
    auto mfma_frag = [&](const Frag &f, auto nr_tag, auto last_tag) {
      constexpr int NR = decltype(nr_tag)::value;
      constexpr bool LAST = decltype(last_tag)::value;
      const bool cok = !(TAIL && LAST) || (cinp - 16 + 4 * g < a.cin);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          const bool ok = ((vmask >> r) & 1u) && cok;
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
#define PCS_PIPE5(LOAD, FR, NRV, LASTV)                                                            \
  LOAD; mfma_frag(FR, std::integral_constant<int, NRV>{}, std::integral_constant<bool, LASTV>{});  \
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
  if (!TAIL || !(nb16 & 1)) { /* an even number of blocks */                                       \
    for (int c0 = 0; c0 < cinp - 32; c0 += 32) {                                                   \
      PCS_PIPE5(load_frag(f1, cur, c0 + 16, true), f0, NRV, false)                                 \
      PCS_PIPE5(load_frag(f0, cur, c0 + 32, true), f1, NRV, false)                                 \
    }                                                                                              \
    PCS_PIPE5(load_frag(f1, cur, cinp - 16, true), f0, NRV, false)                                 \
    make_ctx(nxt, pr_n, vm_n, nr_n, in);                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                             \
    PCS_PIPE5(load_frag(f0, nxt, 0), f1, NRV, true)                                                \
  } else { /* odd: the loop leaves the last block in f0 */                                         \
    for (int c0 = 0; c0 < cinp - 16; c0 += 32) {                                                   \
      PCS_PIPE5(load_frag(f1, cur, c0 + 16, true), f0, NRV, false)                                 \
      PCS_PIPE5(load_frag(f0, cur, c0 + 32, true), f1, NRV, false)                                 \
    }                                                                                              \
    make_ctx(nxt, pr_n, vm_n, nr_n, in);                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                             \
    PCS_PIPE5(load_frag(f1, nxt, 0), f0, NRV, true)                                                \
    f0 = f1;                                                                                       \
  }
    { PCS_BODY5(NRC) }
#undef PCS_BODY5
#undef PCS_PIPE5
    // ---- in-order commit of the group's row blocks -------------------------------------------------------
    // The commits of a workgroup form ONE serial chain (ticket order = group order: bit-reproducible sums); on the
    // sparse full-resolution levels (384-row tiles, 8 waves, ~62 groups per tile) that chain, not the MFMA pipe,
    // bounds the tile. So the row addresses are formed BEFORE the ticket wait, and the wave raises its priority while
    // it holds the ticket (its VALU / LDS instructions otherwise queue behind the MFMA streams of the waves sharing
    // its SIMD): +8 % and +3 % at stride 1. A second ticket for half of the columns bought nothing on top.
    PCS_T(const long long tr_b = wall_clock64();)
    int doff[R][4];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int dl = __shfl(cur.dloc[r], 4 * g + j, 64);
#if PCS_COMMIT_ATOMIC
        doff[r][j] = (dl >= T ? T + g : dl) * C::ACS;  // padding rows: a sink row of this lane group's own (no same-address adds)
#else
        doff[r][j] = dl * C::ACS;
#endif
      }
#if PCS_COMMIT_PHASED && !PCS_COMMIT_ATOMIC
    unsigned dq[R][4], dp[R][4];  // LDS byte addresses of this lane's pieces of the rows it commits
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        dq[r][j] = acc_lds + 4u * (unsigned)doff[r][j] + 16u * l15;
        dp[r][j] = acc_lds + 4u * (unsigned)doff[r][j] + 256u * C::N4 + (C::N2 ? 8u : 4u) * l15;
        asm volatile("" : "+v"(dq[r][j]), "+v"(dp[r][j]));  // formed BEFORE the ticket wait, not sunk into the critical section
      }
#elif PCS_COMMIT_ATOMIC
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(doff[r][j]));
#endif
    if (lane == 0) {
      while (__hip_atomic_load(commit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != grp)
        __builtin_amdgcn_s_sleep(1);
    }
#if PCS_COMMIT_ATOMIC
    // Round 3: the accumulate is ONE ds_add_f32 per lane and tile element, issued while the wave holds the ticket and
    // never waited for. The LDS executes a wave's instructions in order, so the adds of this group reach every address
    // before the ticket store that follows them, and the next owner -- who starts issuing only after it has read the
    // new ticket -- adds after us: the per-address order is the ticket order (bit-reproducible sums, the same fp32
    // additions as the read-add-write form), but the chain link shrinks from [LDS read latency + add + write +
    // completion wait] to the issue time of the adds.
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_setprio(3);
    PCS_T(const long long tr_c = wall_clock64();)
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (r < nr) {  // wave-uniform
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float *d = acc_l + doff[r][j];
#pragma unroll
          for (int q = 0; q < C::N4; ++q) {
            lds_add(d + 64 * q + 4 * l15 + 0, acc[r][4 * q + 0][j]);
            lds_add(d + 64 * q + 4 * l15 + 1, acc[r][4 * q + 1][j]);
            lds_add(d + 64 * q + 4 * l15 + 2, acc[r][4 * q + 2][j]);
            lds_add(d + 64 * q + 4 * l15 + 3, acc[r][4 * q + 3][j]);
          }
          if (C::N2) {
            lds_add(d + 64 * C::N4 + 2 * l15 + 0, acc[r][4 * C::N4 + 0][j]);
            lds_add(d + 64 * C::N4 + 2 * l15 + 1, acc[r][4 * C::N4 + 1][j]);
          }
          if (C::N1) lds_add(d + 64 * C::N4 + 32 * C::N2 + l15, acc[r][NCTT - 1][j]);
        }
      }
    }
#if PCS_COMMIT_NOWAIT
    // the ticket store stays behind the adds in program order and the LDS keeps that order; written as a bare
    // ds_write_b32 -- the compiler puts s_waitcnt lgkmcnt(0) in front of its own store, i.e. the completion wait back
    if (lane == 0) asm volatile("ds_write_b32 %0, %1" ::"v"(commit_lds), "v"(grp + 1) : "memory");
#else
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    if (lane == 0) __hip_atomic_store(commit, grp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
    __builtin_amdgcn_s_setprio(0);
#else
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    __builtin_amdgcn_s_setprio(3);
    PCS_T(const long long tr_c = wall_clock64();)
#if PCS_COMMIT_PHASED
    {
      // Three phases, each behind a compiler barrier: every LDS read of the group (one latency for all of them), every
      // add, every write. Round 2's interleaving went through ~8 read-wait-add rounds per group, each a full LDS
      // latency, inside the one serial chain of the workgroup; the LDS byte addresses (dq: the 16-byte column pieces,
      // dp: the 8-byte pair) are formed before the ticket wait.
      typedef float v2f __attribute__((ext_vector_type(2)));  // native vectors: the HIP float4 / float2 structs do not assign across address spaces
      typedef __attribute__((address_space(3))) const f32x4 lds_cf4;
      typedef __attribute__((address_space(3))) const v2f lds_cf2;
      typedef __attribute__((address_space(3))) const float lds_cf1;
      typedef __attribute__((address_space(3))) f32x4 lds_f4;
      typedef __attribute__((address_space(3))) v2f lds_f2;
      typedef __attribute__((address_space(3))) float lds_f1;
      // all row blocks of the group in one round while the registers allow it (<= 96 columns), else one round per block
      constexpr int RB = (NCTT <= 6) ? NRC : 1;
#pragma unroll
      for (int r0 = 0; r0 < NRC; r0 += RB) {
        f32x4 v4[RB][4][C::N4 > 0 ? C::N4 : 1];
        v2f v2[RB][4];
        float v1[RB][4];
#pragma unroll
        for (int rr = 0; rr < RB; ++rr)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int r = r0 + rr;
#pragma unroll
            for (int q = 0; q < C::N4; ++q) v4[rr][j][q] = *(lds_cf4 *)(size_t)(dq[r][j] + 256u * q);
            if (C::N2) v2[rr][j] = *(lds_cf2 *)(size_t)dp[r][j];
            if (C::N1) v1[rr][j] = *(lds_cf1 *)(size_t)(dp[r][j] + 128u * C::N2);
          }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int rr = 0; rr < RB; ++rr)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int r = r0 + rr;
#pragma unroll
            for (int q = 0; q < C::N4; ++q) {
              v4[rr][j][q].x += acc[r][4 * q + 0][j]; v4[rr][j][q].y += acc[r][4 * q + 1][j];
              v4[rr][j][q].z += acc[r][4 * q + 2][j]; v4[rr][j][q].w += acc[r][4 * q + 3][j];
            }
            if (C::N2) { v2[rr][j].x += acc[r][4 * C::N4 + 0][j]; v2[rr][j].y += acc[r][4 * C::N4 + 1][j]; }
            if (C::N1) v1[rr][j] += acc[r][NCTT - 1][j];
          }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int rr = 0; rr < RB; ++rr)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int r = r0 + rr;
#pragma unroll
            for (int q = 0; q < C::N4; ++q) *(lds_f4 *)(size_t)(dq[r][j] + 256u * q) = v4[rr][j][q];
            if (C::N2) *(lds_f2 *)(size_t)dp[r][j] = v2[rr][j];
            if (C::N1) *(lds_f1 *)(size_t)(dp[r][j] + 128u * C::N2) = v1[rr][j];
          }
        asm volatile("" ::: "memory");
      }
    }
#else
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (r < nr) {  // wave-uniform
        // all LDS reads of the block first (one latency), then the adds, then the writes
        float *d[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) d[j] = acc_l + doff[r][j];
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
#endif
#if PCS_COMMIT_NOWAIT
    // the ticket store stays behind the tile writes in program order and the LDS keeps a wave's instructions in order; a
    // bare ds_write_b32 because the compiler puts the completion wait (s_waitcnt lgkmcnt(0)) in front of its own store
    if (lane == 0) asm volatile("ds_write_b32 %0, %1" ::"v"(commit_lds), "v"(grp + 1) : "memory");
    __builtin_amdgcn_s_setprio(0);
#else
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    if (lane == 0) __hip_atomic_store(commit, grp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __builtin_amdgcn_s_setprio(0);
#endif
#endif
    PCS_T(const long long tr_d = wall_clock64(); tr_loop += tr_b - tr_a; tr_ticket += tr_c - tr_b; tr_commit += tr_d - tr_c; ++tr_groups;)
    cur = nxt;
    i = in;
  };
  {  // wave-uniform loops, no barrier inside: the full groups, then the partial ones (= the commit order)
    int grp = wid;
    for (; grp < total_full; grp += C::NW) run_group(grp, std::integral_constant<int, R>{});
    for (; grp < total_grp; grp += C::NW) run_group(grp, std::integral_constant<int, 1>{});
  }
  PCS_T(const long long tr_end = wall_clock64();)
  __syncthreads();
  const int rows = (int)((a.n_dst - row0) < (int64_t)T ? (a.n_dst - row0) : (int64_t)T);
  float *drow = a.dst + row0 * a.cout + n0;
  const int ldd = a.cout;
  const GStat gstat{a.gs_x, a.gs_mask, a.gs_stat, kGsF32};
  conv_tile_epilogue<C::CT, C::NT>(acc_l, C::ACS, rows, n0, a.cout, a.bias, a.stats ? a.stats + tile * 2 * a.cout : nullptr, tid,
                                   [&](int r, int cq, const float4 &v0) {
                                     float4 v = v0;
                                     if (a.addend) {  // kernel argument: uniform
                                       const float4 ad = *reinterpret_cast<const float4 *>(a.addend + (row0 + r) * (int64_t)ldd + n0 + cq);
                                       v.x += ad.x; v.y += ad.y; v.z += ad.z; v.w += ad.w;
                                     }
                                     if (a.act_slope != 1.f) {
                                       v.x = v.x < 0.f ? v.x * a.act_slope : v.x; v.y = v.y < 0.f ? v.y * a.act_slope : v.y;
                                       v.z = v.z < 0.f ? v.z * a.act_slope : v.z; v.w = v.w < 0.f ? v.w * a.act_slope : v.w;
                                     }
                                     *reinterpret_cast<float4 *>(drow + (int64_t)r * ldd + cq) = v;
                                     return v;
                                   }, a.gs_x ? &gstat : nullptr, row0);
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

template <int NCTT, int NW, int MINW, int R, bool TAIL>
int launch_conv5(const ConvArgs &a, hipStream_t st) {
  using C = Conv5Cfg<NCTT, NW, R>;
  const int64_t nblocks = a.xcd_remap == 2 ? ceil_div(a.ntiles, 8) * 8 * a.ncoltiles : a.ntiles * a.ncoltiles;
  if (nblocks <= 0) return PCS_OK;
  if (nblocks > 0x7FFFFFFF) { set_error("pcs_conv: grid too large"); return PCS_EUNSUPPORTED; }
  auto kern = conv_os5_kernel<NCTT, NW, MINW, R, TAIL>;
  const size_t lds = C::lds_bytes(a.tile_rows);
  if (lds > kMaxDynLds) { set_error("pcs_conv: tile_rows too large for this column tile"); return PCS_EUNSUPPORTED; }
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxDynLds);
    attr_set = true;
  }
  PCS_T(trace_prepare(st);)
  static const int dbg = getenv("PCS_CONV_DEBUG") ? atoi(getenv("PCS_CONV_DEBUG")) : 0;  // debug: launch shape + residency
  static long long dbg_last = -1;
  const long long dbg_key = ((long long)a.tile_rows << 32) ^ ((long long)a.cin << 16) ^ a.cout;
  if (dbg && dbg_key != dbg_last) {
    dbg_last = dbg_key;
    int nb = 0;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void *>(kern), C::NT, lds);
    fprintf(stderr, "[pcs_conv] wave5<%d,%d,%d,%d%s> T=%d cin=%d cout=%d grid=%lld lds=%zu resident WG/CU=%d (waves/SIMD=%d)\n", NCTT, NW,
            MINW, R, TAIL ? ",tail" : "", a.tile_rows, a.cin, a.cout, (long long)nblocks, lds, nb, nb * NW / 4);
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)nblocks), dim3(C::NT), lds, st, a);
  return check_launch("pcs_conv_gather_gemm_f32(wave5)");
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

int pcs::launch_conv_wave5(ConvArgs a, hipStream_t st) {
  int nctt = conv5_nctt(a.cout, a.tile_rows);  // wide outputs on tall tiles: 64-column tiles
  static const int force_nctt = getenv("PCS_CONV_NCTT") ? atoi(getenv("PCS_CONV_NCTT")) : 0;  // debug
  static const int force_nw = getenv("PCS_CONV_NW") ? atoi(getenv("PCS_CONV_NW")) : 0;        // debug: 4 / 8
  if (force_nctt && nctt > force_nctt) nctt = force_nctt;
  a.ncoltiles = (int)ceil_div(a.cout, 16 * nctt);
  // 4-wave workgroups while two of them fit a CU's LDS, else one 8-wave workgroup
  const bool nw8 = force_nw ? force_nw == 8 : 2 * conv5_lds_est(a.tile_rows, nctt) > 160 * 1024;
  // groups of 2 row blocks (groups of 3 / 4 were instantiated and measured through round 2: within +-1 % where they fit
  // the registers -- the W stream they save is not what bounds the kernel -- and spilling at 6 / 8 column tiles)
  const bool tail = (a.cin % 32) != 0;
#define PCS_CONV5_CASE(N)                                                                           \
  case N:                                                                                           \
    if (nw8) return tail ? launch_conv5<N, 8, 2, 2, true>(a, st) : launch_conv5<N, 8, 2, 2, false>(a, st);  \
    return tail ? launch_conv5<N, 4, 2, 2, true>(a, st) : launch_conv5<N, 4, 2, 2, false>(a, st);
  switch (nctt) {
    PCS_CONV5_CASE(2)
    PCS_CONV5_CASE(4)
    PCS_CONV5_CASE(6)
    PCS_CONV5_CASE(8)
  }
#undef PCS_CONV5_CASE
  set_error("pcs_conv_gather_gemm_f32: unreachable");
  return PCS_EINVAL;
}
