// Wave-autonomous fused convolution, one row block per wave step ("wave4"): serves the 16-byte-granular shapes the
// wave5 kernel does not take (cin < 64 or cin % 32 != 0, odd column-tile counts).
#include "conv_common.h"

using namespace pcs;

namespace {

// ================================================================================================
// The block-synchronous kernel (conv_block.hip) spends ~20 scalar+vector instructions per MFMA and half of all wave
// cycles in s_waitcnt / s_barrier (profiles/round1_conv_pmc.md): its stage -> barrier -> MFMA step is too small
// (m ~ 15 rows per offset at 0.05 m LiDAR sparsity). Here the only shared state is the fp32 accumulator tile in LDS:
//   * work item = one 16-row block of ONE offset's compact slice; the waves of a workgroup walk the tile's row
//     blocks round-robin with NO barrier in the main loop;
//   * the wave reads its 16 (src,dst) pairs straight from the rulebook (128 B), gathers its A rows directly in MFMA
//     operand layout (one 16-byte load per lane per 16 channels), and reads the W[k] operand straight from L2 with
//     16-byte loads: lane (g, n) holds W[16j+4g+e][64c+4n .. +3], i.e. B operands of FOUR 16-column tiles whose
//     columns are interleaved (tile f owns columns 4n+f) -- 9 VMEM instructions per 32 MFMAs;
//   * two register sets are software-pipelined (sched_barrier-pinned), the next row block's pairs and first operand
//     block are prefetched across the row-block boundary;
//   * results are committed to the LDS tile in row-block order under an LDS ticket (plain ds_read / add / ds_write:
//     race-free and deterministic; ds_add_f32 measured ~196 cycles per wave instruction);
//   * LDS holds nothing but the accumulator tile -> 3 workgroups per CU at <= 168 VGPRs.
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

  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);  // a scalar: wave-level loops and branches stay uniform
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
  return check_launch("pcs_conv_gather_gemm_f32(wave4)");
}


}  // namespace

int pcs::launch_conv_wave4(ConvArgs a, hipStream_t st) {
  const int nctt = conv_nctt(a.cout);
  a.ncoltiles = (int)ceil_div(a.cout, 16 * nctt);
  // 4 waves with <= 168 VGPRs: 3 workgroups per CU
#define PCS_CONV4_CASE(N)                                                                              \
  case N:                                                                                              \
    return a.tile_rows == 128 ? launch_conv4_cfg<N, 128, 4, 3>(a, st) : launch_conv4_cfg<N, 64, 4, 3>(a, st);
  switch (nctt) {
    PCS_CONV4_CASE(1)
    PCS_CONV4_CASE(2)
    PCS_CONV4_CASE(3)
    PCS_CONV4_CASE(4)
    PCS_CONV4_CASE(6)
    PCS_CONV4_CASE(8)
  }
#undef PCS_CONV4_CASE
  set_error("pcs_conv_gather_gemm_f32: unreachable");
  return PCS_EINVAL;
}
