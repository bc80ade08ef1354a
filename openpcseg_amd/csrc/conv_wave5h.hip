// Half-precision fused convolution (bf16 / fp16 storage, v_mfma_f32_16x16x32_{bf16,f16}, fp32 accumulate): the
// mixed-precision path of the reference (`--amp`: TS:torchsparse/nn/functional/conv.py:19 casts the op inputs to
// half; TS:torchsparse/backend/convolution/convolution_cuda.cu:61,120-127 runs gather / mm / scatter in half).
//
// Same output-stationary structure as conv_wave5.hip -- wave-autonomous row-block groups, fp32 accumulator tile in
// LDS, ticket-ordered commit, every dst row written once -- with the operand side rebuilt for 16-bit MFMAs, whose
// 16x-higher rate turns the kernel from MFMA-bound into L2/gather-bound:
//   * one contraction step = 32 channels = ONE 16-byte load per lane and operand: lane (n = lane & 15, g = lane >> 4)
//     reads the 8 halfs src[row_n][32 s + 8 g .. +7] (A) and the 8 halfs of column n of the weights (B);
//   * the weights are re-packed once per layer call (pcs_conv_prepare_weights_h) into MFMA FRAGMENT ORDER:
//     block (offset k, 16-column tile t, step s) is 1 KB with lane l's 16 bytes at offset 16 l, so a wave's B load
//     is one fully contiguous 1 KB read (eight whole 128-B lines) instead of 16 half-line pieces; the fp32 master
//     weights are converted in the same pass (no separate cast kernel), stored column-major per offset ([column][contraction]) for the forward pass and for dgrad alike;
//   * column tile f of a 64-column quad owns columns 64 q + 4 n + f (as in the fp32 kernel), so the commit and the
//     epilogue move 16-byte LDS words.
// Requires cin % 8 == 0 (16-byte row pieces), cin >= 32, cout % 4 == 0 and an even number of 16-column tiles
// (convh_applies): every layer of the segmentors except the 4/5-channel stems; other shapes are converted to fp32 by
// the host layer and take the fp32 kernels. cin % 32 != 0 (56, 112, 168, 336 of RPVNet cr 1.75 ...) runs the TAIL
// instance: the last step holds 8 / 16 / 24 channels, its out-of-range lane groups read a clamped in-row address
// and are zeroed, and the prepared weights are zero-padded to the full step.
#include "conv_half.h"

using namespace pcs;

namespace pcs {
int launch_conv_wave6h(const ConvArgsH &a, int dtype, hipStream_t st);  // conv_wave6h.hip
bool conv6h_applies(int cin, int cout, int K);
bool conv6h_is_chunked(int cin, int cout);
int conv6h_mode();
}

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));  // native vector: loads from LDS address-space pointers

#if PCS_TRACE
__device__ long long *g_convh_trace;   // [block][wave][8], as conv_wave5.hip: t_entry, t_start, t_end, loop, ticket, commit, groups, t_exit
constexpr int kTraceBlocksH = 8192;
#endif

// Wp block (k, global 16-column tile gt, step s) = 64 lanes x 8 halfs; lane 16 g + n, element j =
//   Wmath[k][32 s + 8 g + j][column(gt, n)],  Wmath[k][c][col] = transpose ? W[k][col][c] : W[k][c][col]
// (W is (K, A, B) fp32: forward contracts over A = cin, dgrad over B = cout). Columns >= ccols are zero.
template <typename HT>
__global__ void __launch_bounds__(256) prepare_weights_kernel(const float *__restrict__ W, int K, int A, int B, int transpose,
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
    const int col = (gt / nctt) * 16 * nctt + h_local_col(nctt, gt % nctt, n);
    uint16_t h[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = 32 * s + 8 * g + j;
      float v = 0.f;
      if (col < ccols && c < ccon)
        v = transpose ? W[((int64_t)k * A + col) * B + c] : W[((int64_t)k * A + c) * B + col];
      h[j] = f2h(HT{}, v);
    }
    uint4 o;
    o.x = h[0] | ((uint32_t)h[1] << 16); o.y = h[2] | ((uint32_t)h[3] << 16);
    o.z = h[4] | ((uint32_t)h[5] << 16); o.w = h[6] | ((uint32_t)h[7] << 16);
    Wp[i] = o;
  }
}

template <int NCTT, int NW_, int R_>
struct Conv5hCfg {
  static constexpr int NW = NW_;
  static constexpr int R = R_;
  static constexpr int NT = 64 * NW;
  static constexpr int CT = 16 * NCTT;
  static constexpr int ACS = CT + 4;
  static constexpr int N4 = NCTT / 4;
  static constexpr int N2 = (NCTT % 4) / 2;
  static constexpr int N1 = NCTT % 2;
  static constexpr int SINK = kConvSinkRows;
  static constexpr size_t lds_bytes(int T) { return (size_t)((T + SINK) * ACS) * 4 + 5 * 33 * 4 + 16; }
};

template <typename HT, int NCTT, int NW, int MINW, int R, bool TAIL>
__global__ void __launch_bounds__(64 * NW, MINW) conv_os5h_kernel(ConvArgsH a) {
  using C = Conv5hCfg<NCTT, NW, R>;
  const int T = a.tile_rows;
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
  if (a.xcd_remap && !a.order) {
    const unsigned nb = gridDim.x, q = nb >> 3, r = nb & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
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
  {  // zero the tile: (T + SINK) * ACS floats, a multiple of four
    float4 *z = reinterpret_cast<float4 *>(acc_l);
    const int n4 = (T + C::SINK) * (C::ACS / 4);
    for (int i = tid; i < n4; i += C::NT) z[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  __syncthreads();
  const int nk = __builtin_amdgcn_readfirstlane(nk_s);  // scalars: the group loop and its branches are wave-uniform
  const int total_full = nk > 0 ? __builtin_amdgcn_readfirstlane(kl_g[nk]) : 0;
  const int total_grp = nk > 0 ? total_full + __builtin_amdgcn_readfirstlane(kl_h[nk]) : 0;

  // B fragments of this column tile: 16-column tiles that do not exist (beyond cout) read tile 0, results dropped
  const int gt0 = ctile * NCTT;
  int btile[NCTT];
#pragma unroll
  for (int t = 0; t < NCTT; ++t) btile[t] = (gt0 + t < a.nt16) ? t : 0;
  const int NS = a.ns;
  // TAIL: channels 32 (NS-1) + 8 g .. +7 of the last step exist only below cin; lane groups beyond it step back to the
  // row's last 8 channels (tail_back bytes) and contribute zeros
  const int tail_over = TAIL ? 32 * (NS - 1) + 8 * g + 8 - a.cin : 0;
  const bool tail_ok = tail_over <= 0;
  const int tail_back = tail_ok ? 0 : 2 * tail_over;

  struct Frag {  // one 32-channel step: A pieces of the R row blocks + the NCTT B fragments
    uint4 a[R];
    uint4 b[NCTT];
  };
  struct Ctx {  // one group: R row blocks of one offset
    const char *srow[R];
    const char *Wk;  // fragment blocks of (offset, first 16-column tile of this column tile), this lane's 16 bytes
    int dloc[R];
    int nr;
    unsigned vmask;
  };
  auto load_frag = [&](Frag &f, const Ctx &cx, int s) {
    const int aoff = TAIL ? s * 64 - (s == NS - 1 ? tail_back : 0) : s * 64;
#if PCS_ABLATEH == 5
#pragma unroll
    for (int r = 0; r < R; ++r) asm volatile("" : "+v"(f.a[r].x), "+v"(f.a[r].y), "+v"(f.a[r].z), "+v"(f.a[r].w));
    (void)aoff;
#else
#pragma unroll
    for (int r = 0; r < R; ++r) f.a[r] = *reinterpret_cast<const uint4 *>(cx.srow[r] + aoff);
#endif
    const bool ldb = PCS_ABLATEH == 1 ? s == 0 : true;   /* ablation 1, timing only: the weight fragments of a group's first step serve all its steps */
#if PCS_ABLATEH == 2 || PCS_ABLATEH == 4   /* timing only: no weight loads at all */
#pragma unroll
    for (int t = 0; t < NCTT; ++t) asm volatile("" : "+v"(f.b[t].x), "+v"(f.b[t].y), "+v"(f.b[t].z), "+v"(f.b[t].w));
#elif PCS_ABLATEH == 6   /* timing only: weight fragments out of LDS (whatever the tile holds), no fill */
#pragma unroll
    for (int t = 0; t < NCTT; ++t)
      f.b[t] = __builtin_bit_cast(uint4, *(__attribute__((address_space(3))) const u32x4 *)(size_t)(acc_lds + (unsigned)((btile[t] * NS + s) * 1024 + lane * 16)));
#else
    if (ldb) {
#pragma unroll
      for (int t = 0; t < NCTT; ++t)
        f.b[t] = *reinterpret_cast<const uint4 *>(cx.Wk + ((size_t)btile[t] * NS + s) * 1024);
    }
#endif
    (void)ldb;
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
      cx.srow[r] = a.src + ((int64_t)(a.src_col ? pr[r].y : pr[r].x) * a.cin + 8 * g) * 2;
      cx.dloc[r] = ((vmask >> r) & 1u) ? (int)((a.src_col ? pr[r].x : pr[r].y) - row0) : T;
    }
    cx.vmask = vmask;
    cx.nr = nr;
    cx.Wk = a.Wp + (((int64_t)kl_k[i_k & 31] * a.nt16 + gt0) * NS) * 1024 + lane * 16;
  };

  int i = 0;
  Ctx cur;
  Frag f0, f1;
#if PCS_ABLATEH
  for (int r = 0; r < R; ++r) f0.a[r] = f1.a[r] = make_uint4(0u, 0u, 0u, 0u);
  for (int t = 0; t < NCTT; ++t) f0.b[t] = f1.b[t] = make_uint4(0u, 0u, 0u, 0u);
#endif
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
  // one straight-line body per group loop (full groups: R row blocks, partial groups: one), see conv_wave5.hip
  static_assert(R >= 2 && R <= 4, "partial groups hold 1 .. R - 1 row blocks");
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
    auto mfma_frag = [&](const Frag &f, auto last_tag) {  // last_tag: the layer's last contraction step
      constexpr bool LAST = decltype(last_tag)::value;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        if (r < nr) {  // wave-uniform
          const bool ok = ((vmask >> r) & 1u) && (!(TAIL && LAST) || tail_ok);
          uint4 av = f.a[r];
          if (!ok) av = make_uint4(0u, 0u, 0u, 0u);  // padding rows contribute exact zeros
#pragma unroll
          for (int t = 0; t < NCTT; ++t) acc[r][t] = mfma_h(HT{}, av, f.b[t], acc[r][t]);
        }
      }
    };
    Ctx nxt;
    // two fragment sets, software-pipelined: the loads of step s+1 are issued before the MFMAs of step s; the first
    // fragment of the NEXT group is in flight during the last step and the commit of this one
    int s = 0;
    for (; s + 2 < NS; s += 2) {
      load_frag(f1, cur, s + 1);
      mfma_frag(f0, std::false_type{});
      load_frag(f0, cur, s + 2);
      mfma_frag(f1, std::false_type{});
    }
    if (NS - s == 2) {
      load_frag(f1, cur, s + 1);
      mfma_frag(f0, std::false_type{});
      make_ctx(nxt, pr_n, vm_n, nr_n, in);
      load_frag(f0, nxt, 0);
      mfma_frag(f1, std::true_type{});
    } else {  // odd number of steps (cin = 96, 160, ...)
      make_ctx(nxt, pr_n, vm_n, nr_n, in);
      load_frag(f1, nxt, 0);
      mfma_frag(f0, std::true_type{});
      f0 = f1;
    }
    PCS_T(const long long tr_b = wall_clock64();)
    // ---- in-order commit of the group's row blocks (as conv_wave5.hip: row addresses formed before the ticket wait,
    // raised wave priority while the ticket is held -- the commits of a workgroup are one serial chain) -----------
    int doff[R][4];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int dl = __shfl(cur.dloc[r], 4 * g + j, 64);
#if PCS_COMMIT_ATOMIC
        doff[r][j] = (dl >= T ? T + g : dl) * C::ACS;  // padding rows: a sink row of this lane group's own
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
#if PCS_ABLATEH == 3 || PCS_ABLATEH == 4   /* timing only: no ticket, no commit */
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int t = 0; t < NCTT; ++t) asm volatile("" ::"v"(acc[r][t]));
    PCS_T(const long long tr_c = wall_clock64();)
#else
    if (lane == 0) {
      while (__hip_atomic_load(commit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != grp)
        __builtin_amdgcn_s_sleep(1);
    }
#if PCS_COMMIT_ATOMIC
    // ds_add_f32 accumulate in ticket order, never waited for (see conv_wave5.hip); columns 16 t + l15: conflict-free rows
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_setprio(3);
    PCS_T(const long long tr_c = wall_clock64();)
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (r < nr) {  // wave-uniform
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float *d = acc_l + doff[r][j] + l15;
#pragma unroll
          for (int t = 0; t < NCTT; ++t) lds_add(d + 16 * t, acc[r][t][j]);
        }
      }
    }
#if PCS_COMMIT_NOWAIT
    if (lane == 0) asm volatile("ds_write_b32 %0, %1" ::"v"(commit_lds), "v"(grp + 1) : "memory");  // see conv_wave5.hip
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
      constexpr int RB = (NCTT <= 6 && NRC >= 2) ? 2 : 1;
#pragma unroll
      for (int r0 = 0; r0 < NRC; r0 += RB) {
        f32x4 v4[RB][4][C::N4 > 0 ? C::N4 : 1];
        v2f v2[RB][4];
        float v1[RB][4];
#pragma unroll
        for (int rr = 0; rr < RB && r0 + rr < NRC; ++rr)
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
        for (int rr = 0; rr < RB && r0 + rr < NRC; ++rr)
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
        for (int rr = 0; rr < RB && r0 + rr < NRC; ++rr)
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
#endif  // PCS_ABLATEH 3 / 4
    PCS_T(const long long tr_d = wall_clock64(); tr_loop += tr_b - tr_a; tr_ticket += tr_c - tr_b; tr_commit += tr_d - tr_c; ++tr_groups;)
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
  PCS_T(const long long tr_end = wall_clock64();)
  __syncthreads();
  // epilogue: fp32 tile (+ fp32 bias) -> halfs, 8-byte stores, every dst row written once
  const int rows = (int)((a.n_dst - row0) < (int64_t)T ? (a.n_dst - row0) : (int64_t)T);
  uint16_t *drow = a.dst + row0 * a.cout + n0;
  const int ldd = a.cout;
  const GStat gstat{a.gs_x, a.gs_mask, a.gs_stat, GsType<HT>::value};
  conv_tile_epilogue<C::CT, C::NT>(acc_l, C::ACS, rows, n0, a.cout, a.bias, a.stats ? a.stats + tile * 2 * a.cout : nullptr, tid,
                                   [&](int r, int cq, const float4 &v0) {
                                     float4 v = v0;
                                     if (a.addend) {  // kernel argument: uniform
                                       const uint2 ad = *reinterpret_cast<const uint2 *>(a.addend + (row0 + r) * (int64_t)ldd + n0 + cq);
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
                                     *reinterpret_cast<uint2 *>(drow + (int64_t)r * ldd + cq) = o;
                                     return make_float4(h2f(HT{}, hx), h2f(HT{}, hy), h2f(HT{}, hz), h2f(HT{}, hw));
                                   }, a.gs_x ? &gstat : nullptr, row0);
#if PCS_TRACE
  if (lane == 0 && blockIdx.x < kTraceBlocksH && g_convh_trace) {
    long long *t = g_convh_trace + ((int64_t)blockIdx.x * 8 + wid) * 8;
    t[0] = tr_entry; t[1] = tr_start; t[2] = tr_end; t[3] = tr_loop; t[4] = tr_ticket; t[5] = tr_commit;
    t[6] = tr_groups; t[7] = wall_clock64();
  }
#endif
}

#if PCS_TRACE
long long *g_traceh_host_ptr = nullptr;
void traceh_prepare(hipStream_t st) {
  if (!g_traceh_host_ptr) {
    (void)hipMalloc(&g_traceh_host_ptr, (size_t)kTraceBlocksH * 64 * sizeof(long long));
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_convh_trace), &g_traceh_host_ptr, sizeof(g_traceh_host_ptr));
  }
  (void)hipMemsetAsync(g_traceh_host_ptr, 0, (size_t)kTraceBlocksH * 64 * sizeof(long long), st);
}
#endif

template <typename HT, int NCTT, int NW, int MINW, int R, bool TAIL>
int launch_conv5h(const ConvArgsH &a, hipStream_t st) {
  using C = Conv5hCfg<NCTT, NW, R>;
  const int64_t nblocks = a.order ? ceil_div(a.ntiles, 8) * 8 * a.ncoltiles : a.ntiles * a.ncoltiles;
  if (nblocks <= 0) return PCS_OK;
  if (nblocks > 0x7FFFFFFF) { set_error("pcs_conv_h: grid too large"); return PCS_EUNSUPPORTED; }
  auto kern = conv_os5h_kernel<HT, NCTT, NW, MINW, R, TAIL>;
  const size_t lds = C::lds_bytes(a.tile_rows);
  if (lds > kMaxDynLds) { set_error("pcs_conv_h: tile_rows too large for this column tile"); return PCS_EUNSUPPORTED; }
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxDynLds);
    attr_set = true;
  }
  PCS_T(traceh_prepare(st);)
  hipLaunchKernelGGL(kern, dim3((unsigned)nblocks), dim3(C::NT), lds, st, a);
  return check_launch("pcs_conv_gather_gemm_h(wave5h)");
}

// row blocks per group of the half kernel per column-tile width (picked from profiles/round3_convh_group_rows.txt)
inline int convh_group_rows(int nctt) {
  (void)nctt;
  return 2;
}

template <typename HT>
int launch_h(ConvArgsH a, hipStream_t st) {
  int nctt = conv_nctt(a.cout);
  static const int force_nctt = getenv("PCS_CONVH_NCTT") ? atoi(getenv("PCS_CONVH_NCTT")) : 0;  // debug: narrower column tiles
  if (force_nctt && nctt > force_nctt && a.cout % (16 * force_nctt) == 0 && !a.stats) nctt = force_nctt;
  a.ncoltiles = (int)ceil_div(a.cout, 16 * nctt);
  const bool nw8 = 2 * conv5_lds_est(a.tile_rows, nctt) > 160 * 1024;  // 4-wave workgroups while two of them fit a CU's LDS, else one 8-wave workgroup
  const bool tail = (a.cin % 32) != 0;
  // row blocks per group: the kernel is bound by the vector-memory address unit (TA ~65-80 % busy, MFMA pipe 10-20 %,
  // profiles/round3_convh_pmc.md): one 16-byte operand load per lane feeds R N / (R + N) MFMAs, so more row blocks per B
  // fragment = fewer loads per MFMA (R = 2, N = 8: 1.6; R = 4, N = 8: 2.67) as far as the registers allow
  static const int force_r = getenv("PCS_CONVH_R") ? atoi(getenv("PCS_CONVH_R")) : 0;  // A/B
  const int rsel = tail ? 2 : (force_r >= 2 && force_r <= 4 ? force_r : convh_group_rows(nctt));
#define PCS_CONV5H_CASE(N)                                                                          \
  case N:                                                                                           \
    if (tail) return nw8 ? launch_conv5h<HT, N, 8, 2, 2, true>(a, st) : launch_conv5h<HT, N, 4, 2, 2, true>(a, st);  \
    if (rsel == 4) return nw8 ? launch_conv5h<HT, N, 8, 2, 4, false>(a, st) : launch_conv5h<HT, N, 4, 2, 4, false>(a, st);  \
    if (rsel == 3) return nw8 ? launch_conv5h<HT, N, 8, 2, 3, false>(a, st) : launch_conv5h<HT, N, 4, 2, 3, false>(a, st);  \
    return nw8 ? launch_conv5h<HT, N, 8, 2, 2, false>(a, st) : launch_conv5h<HT, N, 4, 2, 2, false>(a, st);
  switch (nctt) {
    PCS_CONV5H_CASE(2)
    PCS_CONV5H_CASE(4)
    PCS_CONV5H_CASE(6)
    PCS_CONV5H_CASE(8)
  }
#undef PCS_CONV5H_CASE
  set_error("pcs_conv_gather_gemm_h: unreachable");
  return PCS_EINVAL;
}

}  // namespace

#if PCS_TRACE
// debug builds only: phase timers of the last half-kernel launch (layout as pcs_debug_conv_trace)
extern "C" int pcs_debug_convh_trace(long long *host_out) {
  if (!g_traceh_host_ptr || !host_out) return PCS_EINVAL;
  if (hipDeviceSynchronize() != hipSuccess) return PCS_ELAUNCH;
  return hipMemcpy(host_out, g_traceh_host_ptr, (size_t)kTraceBlocksH * 64 * sizeof(long long), hipMemcpyDeviceToHost) == hipSuccess ? PCS_OK : PCS_ELAUNCH;
}
#endif

extern "C" size_t pcs_conv_prepared_weights_bytes(int32_t K, int32_t ccon, int32_t ccols) {
  if (K <= 0 || ccon <= 0 || ccols <= 0 || ccon % 8) return 0;
  return (size_t)K * (size_t)ceil_div(ccols, 16 * conv_nctt(ccols)) * conv_nctt(ccols) * (size_t)ceil_div(ccon, 32) * 1024;
}

extern "C" int pcs_conv_h_applies(int32_t cin, int32_t cout, int32_t K) { return convh_applies(cin, cout, K) ? 1 : 0; }

extern "C" int pcs_conv_prepare_weights_h(const float *W, int32_t K, int32_t A, int32_t B, int32_t transpose, int32_t dtype,
                                          void *Wp, void *stream) {
  const int ccon = transpose ? B : A, ccols = transpose ? A : B;
  if (K <= 0 || A <= 0 || B <= 0 || !W || !Wp || (dtype != 1 && dtype != 2) || !convh_applies(ccon, ccols, K)) {
    set_error("pcs_conv_prepare_weights_h: bad args / shape not served by the half kernels");
    return PCS_EINVAL;
  }
  const int nctt = conv_nctt(ccols), nt16 = (int)ceil_div(ccols, 16 * nctt) * nctt, ns = (int)ceil_div(ccon, 32);
  const int64_t total = (int64_t)K * nt16 * ns * 64;
  const int grid = stream_grid(total, 256);
  if (dtype == 1)
    hipLaunchKernelGGL(prepare_weights_kernel<Bf16>, dim3(grid), dim3(256), 0, as_stream(stream), W, K, A, B, transpose, nctt,
                       nt16, ns, reinterpret_cast<uint4 *>(Wp));
  else
    hipLaunchKernelGGL(prepare_weights_kernel<Fp16>, dim3(grid), dim3(256), 0, as_stream(stream), W, K, A, B, transpose, nctt,
                       nt16, ns, reinterpret_cast<uint4 *>(Wp));
  return check_launch("pcs_conv_prepare_weights_h");
}

extern "C" int pcs_conv_gather_gemm_h(const void *src, int64_t n_src, int32_t cin, const void *Wp, int32_t K, int32_t cout,
                                      const int32_t *pairs, int32_t src_col, const int32_t *seg, int32_t tile_rows,
                                      int64_t n_dst, const float *bias, void *dst, int32_t dtype, double *bn_partial,
                                      const int32_t *tile_order, void *stream) {
  return pcs_conv_gather_gemm_h_ex(src, n_src, cin, Wp, K, cout, pairs, src_col, seg, tile_rows, n_dst, bias, nullptr, dst, dtype,
                                   bn_partial, tile_order, stream);
}

extern "C" int pcs_conv_gather_gemm_h_ex(const void *src, int64_t n_src, int32_t cin, const void *Wp, int32_t K, int32_t cout,
                                         const int32_t *pairs, int32_t src_col, const int32_t *seg, int32_t tile_rows,
                                         int64_t n_dst, const float *bias, const pcs_conv_epilogue *ep, void *dst, int32_t dtype,
                                         double *bn_partial, const int32_t *tile_order, void *stream) {
  const void *addend = ep ? ep->addend : nullptr;
  if (cin <= 0 || cout <= 0 || K <= 0 || n_dst < 0 || n_src < 0 || (src_col != 0 && src_col != 1) || (dtype != 1 && dtype != 2)) {
    set_error("pcs_conv_gather_gemm_h: bad sizes");
    return PCS_EINVAL;
  }
  if (!convh_applies(cin, cout, K)) { set_error("pcs_conv_gather_gemm_h: shape not served by the half kernels (needs cin %% 8 == 0, cin >= 32, cout %% 4 == 0, an even number of 16-column tiles)"); return PCS_EUNSUPPORTED; }
  if (n_dst == 0) return PCS_OK;
  if (!Wp || !seg || !dst || (n_src > 0 && !src)) { set_error("pcs_conv_gather_gemm_h: null pointer"); return PCS_EINVAL; }
  if ((((uintptr_t)src | (uintptr_t)Wp | (uintptr_t)bias) & 15) || ((uintptr_t)dst & 7)) { set_error("pcs_conv_gather_gemm_h: misaligned pointer"); return PCS_EINVAL; }
  if (tile_rows < 16 || tile_rows > 512 || tile_rows % 16) { set_error("pcs_conv_gather_gemm_h: tile_rows must be a multiple of 16 in [16, 512]"); return PCS_EINVAL; }
  ConvArgsH a;
  a.src = reinterpret_cast<const char *>(src); a.Wp = reinterpret_cast<const char *>(Wp); a.bias = bias;
  a.dst = reinterpret_cast<uint16_t *>(dst); a.pairs = pairs; a.seg = seg;
  a.n_dst = n_dst; a.ntiles = ceil_div(n_dst, tile_rows); a.tile_rows = tile_rows;
  a.cin = cin; a.cout = cout; a.K = K; a.src_col = src_col; a.ncoltiles = 1; a.xcd_remap = 1; a.stats = bn_partial; a.order = tile_order;
  a.addend = reinterpret_cast<const uint16_t *>(addend);
  if (addend && ((uintptr_t)addend & 7)) { set_error("pcs_conv_gather_gemm_h_ex: misaligned addend"); return PCS_EINVAL; }
  if (ep && ep->bn_x) {
    if (!bn_partial || !ep->bn_stat || ((uintptr_t)ep->bn_x & 7) || (ep->bn_mask && (cout & 31))) {
      set_error("pcs_conv_gather_gemm_h_ex: BatchNorm backward statistics need bn_partial, bn_stat, aligned bn_x and cout %% 32 == 0 with a gate mask");
      return PCS_EINVAL;
    }
    a.gs_x = ep->bn_x; a.gs_mask = ep->bn_mask; a.gs_stat = ep->bn_stat;
  }
  if (ep && ep->act_slope != 0.f && ep->act_slope != 1.f) a.act_slope = ep->act_slope;
  if (bn_partial && !pcs_conv_emits_bn_partials(cin, cout, K, tile_rows, dtype)) {
    set_error("pcs_conv_gather_gemm_h: this shape / tile height does not produce BatchNorm partials");
    return PCS_EUNSUPPORTED;
  }
  const int nctt = conv_nctt(cout);
  a.nt16 = (int)ceil_div(cout, 16 * nctt) * nctt;
  a.ns = (int)ceil_div(cin, 32);
  // the weight-stationary kernel (conv_wave6h.hip); its chunked-contraction instances only on 4-wave workgroups (<= 160-row tiles)
  if (conv6h_mode() && conv6h_applies(cin, cout, K) && (!conv6h_is_chunked(cin, cout) || tile_rows <= 160))
    return launch_conv_wave6h(a, dtype, as_stream(stream));
  return dtype == 1 ? launch_h<Bf16>(a, as_stream(stream)) : launch_h<Fp16>(a, as_stream(stream));
}
