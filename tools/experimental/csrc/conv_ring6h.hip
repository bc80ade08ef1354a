// Half-precision fused convolution, column-parallel ("ring") form: bf16 / fp16 storage, v_mfma_f32_16x16x32_{bf16,f16},
// fp32 accumulate. Replaces the same reference dataflow as conv_wave5h.hip (gather -> mm -> scatter per offset,
// TS:torchsparse/backend/convolution/convolution_cuda.cu:53-165 under `--amp`) for the 96 / 128-column tiles of the
// >= 64-channel layers.
//
// Why a second structure. conv_wave5h.hip hands row-block GROUPS of one offset to the waves of a workgroup; every wave
// streams the weight slab W[k] of its group out of L2 again, so 75-80 % of the kernel's vector loads are private copies
// of the same B fragments and the vector-memory address path (TA), not the MFMA pipe, is what saturates
// (profiles/round3_convh_pmc.md). Here the waves of a workgroup split the tile's COLUMNS instead of its row blocks:
//   * compute wave w owns NC 16-column tiles of the CT-wide tile and walks ALL row blocks of ALL offsets of the tile
//     in rulebook order. Its B fragments (the slab W[k][chunk of KC 32-channel steps][its columns]) are loaded ONCE per
//     (tile, offset, chunk) straight into registers -- three rotating register sets, loaded two jobs ahead -- so every
//     weight byte enters the CU once per tile;
//   * the gathered A rows are needed by every compute wave: a LOADER wave brings them in once, by LDS-DMA
//     (global_load_lds_dwordx4: per-lane gather address, 1 KB per instruction landing in MFMA fragment order), into a
//     ring of kRingDepth batches (batch = 2 row blocks x KC steps); the compute waves read the fragments with
//     ds_read_b128 (conflict-free, 256 B/clk/CU against the TA's 64). The loader reads the pair indices through the
//     same path (global_load_lds_dword into a small index ring), so it has no register-destination loads at all and
//     counts its own vmcnt; one s_barrier per batch hands a landed batch to the compute waves and a consumed slot back;
//   * waves own disjoint columns of the fp32 accumulator tile in LDS and commit in program order: no ticket, no
//     atomics, bit-reproducible; the commit of a row block is issued behind the MFMAs of the next one.
// The tile is as tall as the LDS allows beside the ring (one workgroup per CU: W is amortised over more rows);
// epilogue, BatchNorm partials, prepared-weight format, tile order and XCD mapping are shared with conv_wave5h.hip.
#include "conv_ring.h"

using namespace pcs;

namespace {

typedef ring_v2f v2f;
typedef ring_v4i v4i;
typedef ring_v4u v4u;

template <int NCTT_, int NC_, int KC_>
struct RingCfg {
  static constexpr int NCTT = NCTT_, NC = NC_, KC = KC_;
  static constexpr int NWC = NCTT / NC;   // compute waves
  static constexpr int NW = NWC + 1;      // + the loader
  static constexpr int NT = 64 * NW;
  static constexpr int CT = 16 * NCTT;
  static constexpr int ACS = CT + 4;
  static constexpr int BR = kRingBatchRows;
  static constexpr int D = kRingDepth;
  static constexpr int MD = kRingMeta;
  static constexpr int BK = BR * KC;      // 1 KB fragments per batch
  static constexpr int PF = BK >= 6 ? 4 : 2;  // fragments read ahead of the MFMAs
  static constexpr int RING_BYTES = D * BK * 1024;
  static constexpr int META_BYTES = MD * BR * 128;
  static_assert(MD >= 2 * D - 3 && (MD & (MD - 1)) == 0, "index ring too short");
  static_assert((D - 3) * (1 + BK) <= 63, "vmcnt field");
  static_assert(NC == 1 || NC == 2, "a compute wave owns one or two 16-column tiles");
};

template <typename HT, int NCTT, int NC, int KC>
__global__ void __launch_bounds__(64 * (NCTT / NC + 1), 1) conv_ring6h_kernel(ConvArgsH a) {
  using C = RingCfg<NCTT, NC, KC>;
  constexpr int BR = C::BR, D = C::D, MD = C::MD, BK = C::BK, PF = C::PF;
  const int T = a.tile_rows;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // [A ring | pair-index ring | offset lists | batch table | accumulator tile]: the DMA targets come first (low LDS addresses)
  int *meta = reinterpret_cast<int *>(smem + C::RING_BYTES);           // [MD][BR][2][16]: src rows, dst rows of a batch
  int *kl_k = meta + C::META_BYTES / 4;                                // [36] offset id
  int *kl_s = kl_k + 36;                                               // [36] first pair
  int *kl_m = kl_s + 36;                                               // [36] #pairs
  int *kl_b = kl_m + 36;                                               // [36] first batch (prefix over the offsets)
  int *misc = kl_b + 36;                                               // nk, NB
  int2 *bt = reinterpret_cast<int2 *>(misc + 4);                       // [NB] batch -> {first pair, pairs left in the slice (<= 255) | chunk << 8}
  float *acc_l = reinterpret_cast<float *>(smem + a.ring_acc_off);     // [T+1][ACS], row T = sink of the padding rows
  const unsigned ring_lds = (unsigned)(size_t)(PCS_LDS(char) *)smem;
  const unsigned meta_lds = (unsigned)(size_t)(PCS_LDS(int) *)meta;
  const unsigned acc_lds = (unsigned)(size_t)(PCS_LDS(float) *)acc_l;

  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, l15 = lane & 15;
  unsigned bid = blockIdx.x;
  if (a.xcd_remap && !a.order) {
    const unsigned nb = gridDim.x, q = nb >> 3, r = nb & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  int64_t tslot = bid / a.ncoltiles;
  int ctile = bid % a.ncoltiles;
  if (a.order) {  // tiles dealt round-robin over the XCDs, the column tiles of one row tile back to back on one XCD
    const unsigned xcd = bid & 7, idx = bid >> 3;
    tslot = (int64_t)(idx / a.ncoltiles) * 8 + xcd;
    ctile = idx % a.ncoltiles;
    if (tslot >= a.ntiles) return;  // the grid is padded to 8 * ncoltiles
  }
  const int64_t tile = a.order ? (int64_t)a.order[tslot] : tslot;
  const int n0 = ctile * C::CT;
  const int64_t row0 = tile * T;
  const int64_t nt1 = a.ntiles + 1;
  const int NS = a.ns, NCH = NS / KC;

  if (wid == 0) {  // non-empty offsets of this tile + prefix of their batches
    const int k = lane;
    int s0 = 0, m = 0;
    if (k < a.K) {
      s0 = a.seg[(int64_t)k * nt1 + tile];
      m = a.seg[(int64_t)k * nt1 + tile + 1] - s0;
    }
    const unsigned long long mask = __ballot(m > 0);
    const int nrb = (m + 15) >> 4;
    const int nbat = NCH * ((nrb + BR - 1) / BR);
    int incl = nbat;
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(incl, o, 64);
      if (lane >= o) incl += t;
    }
    if (m > 0) {
      const int pos = __popcll(mask & ((1ULL << lane) - 1ULL));
      kl_k[pos] = k; kl_s[pos] = s0; kl_m[pos] = m; kl_b[pos] = incl - nbat;
    }
    const int total = __shfl(incl, 63, 64);
    if (lane == 0) {
      const int nkk = __popcll(mask);
      misc[0] = nkk; misc[1] = total; kl_b[nkk] = total;
    }
  }
  {  // zero the tile: (T + 1) * ACS floats, a multiple of four
    float4 *z = reinterpret_cast<float4 *>(acc_l);
    const int n4 = (T + 1) * (C::ACS / 4);
    for (int i = tid; i < n4; i += C::NT) z[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  __syncthreads();
  const int nk = __builtin_amdgcn_readfirstlane(misc[0]);
  const int NB = __builtin_amdgcn_readfirstlane(misc[1]);
  for (int e = 0; e < nk; ++e) {  // batch table: job-major ((offset, chunk) outer, row blocks inner)
    const int m = kl_m[e], s0 = kl_s[e], nrb = (m + 15) >> 4, nbe = (nrb + BR - 1) / BR, base = kl_b[e];
    for (int idx = tid; idx < NCH * nbe; idx += C::NT) {
      const int c = idx / nbe, bb = idx - c * nbe, left = m - bb * BR * 16;
      bt[base + idx] = make_int2(s0 + bb * BR * 16, (left < 255 ? left : 255) | (c << 8));
    }
  }
  __syncthreads();

  if (NB > 0) {
    if (wid == C::NWC) {
      // ---------------- loader: pair indices and gathered rows by LDS-DMA, D - 2 batches ahead of the compute waves -------------
      // Iteration i (after barrier i, which tells that every compute wave has consumed batch i - 2) issues the index DMA of
      // batch i + LP and the row DMAs of batch i + LA into the freed slot, then waits until only the DMAs of the last W
      // iterations are outstanding: batch i + 1 has landed before barrier i + 1. Everything an iteration needs from LDS (its
      // two batch descriptors, the 2 x 16 source rows of its row DMA) is read one iteration ahead, behind the previous
      // iteration's DMA issue: no LDS round trip sits between a barrier and the first DMA (the first version chained four
      // of them there and ran at ~1700 cycles per batch whatever the ring depth, profiles/round4_ring.md).
      constexpr int LA = D - 2, LP = 2 * D - 3, W = D - 3;
      static_assert(MD >= LP + 1, "index ring too short");
      const int prow = (lane >> 5) * 16 + l15;                           // index DMA: lanes 0-31 row block 0, 32-63 row block 1
      const int pcol = ((lane >> 4) & 1) ? 1 - a.src_col : a.src_col;     // 16 src rows then 16 dst rows per row block
      const int32_t *pbase = a.pairs + pcol;
      const char *abase = a.src + 16 * g;
      const int64_t row_bytes = (int64_t)a.cin * 2;
      auto desc = [&](int x) { return bt[x < NB ? x : NB - 1]; };        // same value in every lane
      auto idx_of = [&](int x, int r) { return meta[(x & (MD - 1)) * (BR * 32) + r * 32 + l15]; };
      auto issue_P = [&](int2 d, int x) {  // one DMA: the (src, dst) rows of the batch's 2 x 16 pairs; padding rows re-read the slice's last pair
        const int first = __builtin_amdgcn_readfirstlane(d.x), left = __builtin_amdgcn_readfirstlane(d.y) & 255;
        const int rk = prow < left ? prow : left - 1;
        glds4(pbase + (int64_t)(first + rk) * 2, __builtin_amdgcn_readfirstlane(meta_lds + (unsigned)(x & (MD - 1)) * (BR * 128)));
      };
      auto issue_A = [&](int2 d, const int (&idx)[BR], int slot) {  // BR * KC DMAs of 1 KB: fragment (r, s) = rows of row block r, channels 32 (c KC + s) + 8 g .. + 7
        const int c = __builtin_amdgcn_readfirstlane(d.y) >> 8;
        const unsigned dst = __builtin_amdgcn_readfirstlane(ring_lds + (unsigned)(slot * BK * 1024));
#pragma unroll
        for (int r = 0; r < BR; ++r) {
          const char *base = abase + (int64_t)idx[r] * row_bytes + c * (KC * 64);
#pragma unroll
          for (int s = 0; s < KC; ++s) {
            if (PCS_RING_ABLATE == 2) glds4(a.pairs + lane, dst + (unsigned)((r * KC + s) * 1024));  // same op count, no row traffic
            else glds16(base + s * 64, dst + (unsigned)((r * KC + s) * 1024));
          }
        }
      };
#pragma unroll
      for (int x = 0; x < LP - LA; ++x) issue_P(desc(x), x);             // the indices the first LA iterations' row DMAs read
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      int2 dA = desc(0), dP = desc(LP - LA);
      int idx[BR];
#pragma unroll
      for (int r = 0; r < BR; ++r) idx[r] = idx_of(0, r);
      int slot = 0;
      for (int i = -LA; i < NB; ++i) {
        if (i >= 0) ring_barrier();
        // next iteration's operands: their indices landed an iteration ago (LP - LA = W + 2 iterations between an index DMA and
        // the read of what it brought)
        const int2 dA_n = desc(i + 1 + LA), dP_n = desc(i + 1 + LP);
        int idx_n[BR];
#pragma unroll
        for (int r = 0; r < BR; ++r) idx_n[r] = idx_of(i + 1 + LA, r);
        if (i + LA < NB) {
          issue_P(dP, i + LP);
          if (PCS_RING_ABLATE != 5) issue_A(dA, idx, slot);
          slot = slot + 1 == D ? 0 : slot + 1;
          if (PCS_RING_ABLATE != 6 && PCS_RING_ABLATE != 5) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(W * (1 + BK)) : "memory");
          if (PCS_RING_ABLATE == 5) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(W) : "memory");
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        dA = dA_n; dP = dP_n;
#pragma unroll
        for (int r = 0; r < BR; ++r) idx[r] = idx_n[r];
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // nothing of this workgroup's DMA may land after its LDS is released
    } else {
      // ---------------- compute wave: NC 16-column tiles, every row block of every offset ------------------------------------
      const int gt0 = ctile * NCTT;
      int wtile[NC];  // 16-column tiles that do not exist (beyond cout) read tile 0, results dropped by the epilogue
#pragma unroll
      for (int tt = 0; tt < NC; ++tt) wtile[tt] = (gt0 + wid * NC + tt < a.nt16) ? wid * NC + tt : 0;
      const unsigned col_off = 4u * (unsigned)h_local_col(NCTT, wid * NC, l15);  // NC = 2: columns (col, col + 1) of the pair of tiles
      const int NJ = nk * NCH;
      uint4 B[3][NC][KC];
      auto load_B = [&](auto set_tag, int e, int c) {
        constexpr int S = decltype(set_tag)::value;
        const int k = PCS_RING_ABLATE == 1 ? 0 : __builtin_amdgcn_readfirstlane(kl_k[e]);
        if (PCS_RING_ABLATE == 1 && e > 0) return;
#pragma unroll
        for (int tt = 0; tt < NC; ++tt)
#pragma unroll
          for (int s = 0; s < KC; ++s)
            B[S][tt][s] = *reinterpret_cast<const uint4 *>(a.Wp + (((int64_t)k * a.nt16 + gt0 + wtile[tt]) * NS + c * KC + s) * 1024 + lane * 16);
      };
      auto read_frag = [&](int slot, int f) {
        const v4u v = *(PCS_LDS(const v4u) *)(size_t)(ring_lds + (unsigned)((slot * BK + f) * 1024 + lane * 16));
        return make_uint4(v.x, v.y, v.z, v.w);
      };
      auto commit = [&](const f32x4 (&ac)[NC], const unsigned (&rw)[4]) {  // this wave's columns of 16 dst rows: read, add, write
        if (PCS_RING_ABLATE == 3) {
          if (ac[0][0] == 12345.678f) *(PCS_LDS(float) *)(size_t)rw[0] = ac[0][1];
          return;
        }
        if constexpr (NC == 2) {
          v2f v[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = *(PCS_LDS(const v2f) *)(size_t)rw[j];
#pragma unroll
          for (int j = 0; j < 4; ++j) { v[j].x += ac[0][j]; v[j].y += ac[1][j]; }
#pragma unroll
          for (int j = 0; j < 4; ++j) *(PCS_LDS(v2f) *)(size_t)rw[j] = v[j];
        } else {
          float v[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = *(PCS_LDS(const float) *)(size_t)rw[j];
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] += ac[0][j];
#pragma unroll
          for (int j = 0; j < 4; ++j) *(PCS_LDS(float) *)(size_t)rw[j] = v[j];
        }
      };

      int i = 0, slot = 0;               // batch, its ring slot
      int je = 0, jc = 0;                // (offset, chunk) of the running job
      int pe = 0, pc = 0, pj = 0;        // job whose weights are loaded next (clamped to the last one)
      auto advance = [&](int &e, int &c) { if (++c == NCH) { c = 0; ++e; } };
      uint4 afr[BK];
      f32x4 acc1p[NC];                   // row block 1 of the previous batch: committed behind the first MFMAs of the next
      unsigned rows1p[4];
      bool havep = false;
#pragma unroll
      for (int tt = 0; tt < NC; ++tt) acc1p[tt] = (f32x4){0, 0, 0, 0};
#pragma unroll
      for (int j = 0; j < 4; ++j) rows1p[j] = acc_lds;

      // Weight loads are UNCONDITIONAL (past the last job they re-read its slab): with a conditional prefetch the compiler's
      // vmcnt for the running job's fragments must also hold on the path that issued nothing behind them, and every job
      // would drain the prefetch of the next two.
      auto next_job = [&]() { if (pj + 1 < NJ) advance(pe, pc); ++pj; };
      load_B(std::integral_constant<int, 0>{}, pe, pc); next_job();
      load_B(std::integral_constant<int, 1>{}, pe, pc); next_job();
      ring_barrier();  // barrier 0: batch 0 has landed
#pragma unroll
      for (int f = 0; f < PF; ++f) afr[f] = read_frag(0, f);

      auto body = [&](auto set_tag, int rb0, int m) {
        constexpr int S = decltype(set_tag)::value;
        if (PCS_RING_ABLATE == 7) {  // barriers only
          if (i + 1 < NB) ring_barrier();
          ++i;
          if (rb0 == 12345) acc1p[0] = mfma_h(HT{}, afr[0], B[S][0][0], acc1p[0]);
          return;
        }
        unsigned rows[BR][4];
#pragma unroll
        for (int r = 0; r < BR; ++r) {
          const v4i d4 = *(PCS_LDS(const v4i) *)(size_t)(meta_lds + (unsigned)((i & (MD - 1)) * (BR * 128) + r * 128 + 64 + g * 16));
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int rk = (rb0 + r) * 16 + 4 * g + j;
            const int dl = rk < m ? d4[j] - (int)row0 : T;   // padding rows (and the row block a partial batch lacks) go to the sink row
            rows[r][j] = acc_lds + __umul24((unsigned)dl, (unsigned)(C::ACS * 4)) + col_off;
          }
        }
        const bool more = i + 1 < NB;
        const int nslot = slot + 1 == D ? 0 : slot + 1;
        f32x4 acc[BR][NC];
#pragma unroll
        for (int r = 0; r < BR; ++r)
#pragma unroll
          for (int tt = 0; tt < NC; ++tt) acc[r][tt] = (f32x4){0, 0, 0, 0};
#pragma unroll
        for (int f = 0; f < BK; ++f) {
          const int r = f / KC, s = f % KC;
          if (f == BK - PF && more) ring_barrier();  // barrier i + 1: the next batch has landed; this wave is done with batch i - 1
          if (f + PF < BK) afr[f + PF] = read_frag(slot, f + PF);
          else if (more) afr[f + PF - BK] = read_frag(nslot, f + PF - BK);
#pragma unroll
          for (int tt = 0; tt < NC; ++tt) {
            if (PCS_RING_ABLATE == 4) { acc[r][tt][0] += __uint_as_float(afr[f].x ^ B[S][tt][s].y); }
            else acc[r][tt] = mfma_h(HT{}, afr[f], B[S][tt][s], acc[r][tt]);
          }
          if (f == 1 && havep) commit(acc1p, rows1p);
          if (f == KC + 1) commit(acc[0], rows[0]);
        }
        if (KC == 1) commit(acc[0], rows[0]);
#pragma unroll
        for (int tt = 0; tt < NC; ++tt) acc1p[tt] = acc[1][tt];
#pragma unroll
        for (int j = 0; j < 4; ++j) rows1p[j] = rows[1][j];
        havep = true;
        ++i;
        slot = nslot;
      };
      auto run_job = [&](auto set_tag) {  // one (offset, chunk): its weights are in register set S; load the job after next into S + 2
        constexpr int S = decltype(set_tag)::value;
        load_B(std::integral_constant<int, (S + 2) % 3>{}, pe, pc); next_job();
        const int m = __builtin_amdgcn_readfirstlane(kl_m[je]);
        const int nrb = (m + 15) >> 4;
        int rb0 = 0;  // m > 0: at least one batch (a do-while: the weights of set S are provably consumed before S is loaded again)
        do { body(set_tag, rb0, m); rb0 += BR; } while (rb0 < nrb);
        advance(je, jc);
      };
      // jobs in threes (the register sets rotate statically), one loop exit; the one or two jobs left over run behind it
      for (int jg = NJ / 3; jg > 0; --jg) {
        run_job(std::integral_constant<int, 0>{});
        run_job(std::integral_constant<int, 1>{});
        run_job(std::integral_constant<int, 2>{});
      }
      const int jrem = NJ % 3;
      if (jrem >= 1) run_job(std::integral_constant<int, 0>{});
      if (jrem == 2) run_job(std::integral_constant<int, 1>{});
      if (havep) commit(acc1p, rows1p);
    }
  }
  __syncthreads();
  // epilogue: fp32 tile (+ fp32 bias) -> halfs, 8-byte stores, every dst row written once (shared with conv_wave5h.hip)
  const int rows = (int)((a.n_dst - row0) < (int64_t)T ? (a.n_dst - row0) : (int64_t)T);
  uint16_t *drow = a.dst + row0 * a.cout + n0;
  const int ldd = a.cout;
  conv_tile_epilogue<C::CT, C::NT>(acc_l, C::ACS, rows, n0, a.cout, a.bias, a.stats ? a.stats + tile * 2 * a.cout : nullptr, tid,
                                   [&](int r, int cq, const float4 &v) {
                                     const uint16_t hx = f2h(HT{}, v.x), hy = f2h(HT{}, v.y), hz = f2h(HT{}, v.z), hw = f2h(HT{}, v.w);
                                     uint2 o;
                                     o.x = hx | ((uint32_t)hy << 16);
                                     o.y = hz | ((uint32_t)hw << 16);
                                     *reinterpret_cast<uint2 *>(drow + (int64_t)r * ldd + cq) = o;
                                     return make_float4(h2f(HT{}, hx), h2f(HT{}, hy), h2f(HT{}, hz), h2f(HT{}, hw));
                                   });
}

template <typename HT, int NCTT, int NC, int KC>
int launch_ring(const ConvArgsH &a, size_t lds, hipStream_t st) {
  using C = RingCfg<NCTT, NC, KC>;
  const int64_t nblocks = a.order ? ceil_div(a.ntiles, 8) * 8 * a.ncoltiles : a.ntiles * a.ncoltiles;
  if (nblocks <= 0) return PCS_OK;
  if (nblocks > 0x7FFFFFFF) { set_error("pcs_conv_h: grid too large"); return PCS_EUNSUPPORTED; }
  auto kern = conv_ring6h_kernel<HT, NCTT, NC, KC>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxDynLds);
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)nblocks), dim3(C::NT), lds, st, a);
  return check_launch("pcs_conv_gather_gemm_h(ring6h)");
}

template <typename HT>
int launch_ring_dt(const ConvArgsH &a, const RingShape &s, size_t lds, hipStream_t st) {
#define PCS_RING_KC(N, C_)                                                  \
  switch (s.kc) {                                                           \
    case 2: return launch_ring<HT, N, C_, 2>(a, lds, st);                   \
    case 3: return launch_ring<HT, N, C_, 3>(a, lds, st);                   \
    case 4: return launch_ring<HT, N, C_, 4>(a, lds, st);                   \
  }
  if (s.nctt == 6 && s.nc == 2) { PCS_RING_KC(6, 2) }
  if (s.nctt == 8 && s.nc == 2) { PCS_RING_KC(8, 2) }
  if (s.nctt == 6 && s.nc == 1) { PCS_RING_KC(6, 1) }
  if (s.nctt == 8 && s.nc == 1) { PCS_RING_KC(8, 1) }
#undef PCS_RING_KC
  set_error("pcs_conv_gather_gemm_h(ring6h): unreachable");
  return PCS_EINVAL;
}

}  // namespace

namespace pcs {

int &conv_ring_mode() {
  static int mode = getenv("PCS_CONVH_RING") ? atoi(getenv("PCS_CONVH_RING")) : -1;
  return mode;
}

int launch_conv_ring6h(const ConvArgsH &a0, int dtype, hipStream_t st) {
  RingShape s;
  if (!conv_ring_applies(a0.cin, a0.cout, a0.K, a0.tile_rows, &s)) { set_error("pcs_conv_gather_gemm_h(ring6h): shape not served"); return PCS_EUNSUPPORTED; }
  ConvArgsH a = a0;
  a.ncoltiles = (int)ceil_div(a.cout, 16 * s.nctt);
  int acc_off = 0;
  const size_t lds = conv_ring_lds(a.tile_rows, s, a.ns, a.K, &acc_off);
  a.ring_acc_off = acc_off;
  a.ring_bt_cap = conv_ring_bt_cap(a.tile_rows, a.ns, s.kc, a.K);
  return dtype == 1 ? launch_ring_dt<Bf16>(a, s, lds, st) : launch_ring_dt<Fp16>(a, s, lds, st);
}

}  // namespace pcs

