// fp32 fused convolution, column-parallel ("ring") form: fp32 storage, v_mfma_f32_16x16x4_f32 (exact fp32 products,
// fp32 accumulate). Replaces the same reference dataflow as conv_wave5.hip (gather -> mm -> scatter per offset,
// TS:torchsparse/backend/convolution/convolution_cuda.cu:53-165) for layers with cin % 32 == 0 and 64 / 96 / 128-column
// output tiles.
//
// Structure = conv_ring6h.hip: the compute waves of a workgroup split the tile's COLUMNS (one or two 16-column MFMA
// tiles each) and walk every row block of every offset in rulebook order; a loader wave gathers the A rows once per
// column tile by LDS-DMA into a ring (fragment = 16 rows x 16 channels = 1 KB, lane (n, g) holds channels 4 g .. 4 g + 3
// of row n: the operand layout of four v_mfma_f32_16x16x4_f32); waves own disjoint columns of the fp32 accumulator tile
// in LDS and commit in program order (no ticket, no atomics, bit-reproducible).
// Why it pays HERE and not for the 16-bit kernels (profiles/round4_ring.md): the fp32 MFMA issues once per 32 cycles,
// so a 16-channel fragment carries 4 x NC x 32 = 256 cycles of matrix-pipe time per wave against ~40 other
// instructions per row block (ring reads, row addresses, commit) -- the per-row-block bookkeeping that every column
// wave repeats, and that left the bf16 form behind conv_os5h_kernel, hides under the MFMAs. What the wave-autonomous
// kernel loses (ticket order waits, serial prologue of 2-4 workgroups per CU, groups of uneven length at the tail of a
// tile) has no counterpart here: every compute wave does the same work on every row block.
// The B operand is read straight from the (K, cin, cout) weights (8-byte pieces: the two adjacent columns of the wave's
// tile pair), once per (tile, offset, chunk of KC 32-channel steps), two register sets, loaded one job ahead.
#include "conv_ring.h"

using namespace pcs;

namespace {

typedef ring_v2f v2f;
typedef ring_v4i v4i;
typedef float v4f __attribute__((ext_vector_type(4)));

#ifndef PCS_RINGF_ABLATE
#define PCS_RINGF_ABLATE 0   /* variant builds, bit mask: 1 no commits, 2 no ring reads, 4 no MFMA, 8 no row addresses, 16 no weight loads after the first job,
                                32 no row DMA, 64 commits without their LDS reads, 128 commits without their LDS writes,
                                512 no MFMA / side-work interleave */
#endif
#ifndef PCS_RING_TRACE
#define PCS_RING_TRACE 0   /* variant builds: per-wave phase timers (s_memtime) of the first 1024 workgroups, read with pcs_debug_ringf_trace */
#endif
#if PCS_RING_TRACE
__device__ long long *g_ringf_trace;   // [block][wave][8]: entry, main start, main end, exit, barrier cycles, DMA wait / job-start cycles, issue cycles, batches
constexpr int kRingTraceBlocks = 1024;
#define PCS_RT(...) __VA_ARGS__
#else
#define PCS_RT(...)
#endif

template <int NCTT_, int NC_, int KC_>
struct RingFCfg {
  static constexpr int NCTT = NCTT_, NC = NC_, KC = KC_;
  static constexpr int NWC = NCTT / NC;   // compute waves
  static constexpr int NW = NWC + 1;      // + the loader
  static constexpr int NT = 64 * NW;
  static constexpr int CT = 16 * NCTT;
  static constexpr int ACS = CT + 4;
  static constexpr int BR = kRingBatchRows;
  static constexpr int D = kRingDepth;
  static constexpr int MD = kRingMeta;
  static constexpr int FK = 2 * KC;       // 16-channel fragments per row block and chunk
  static constexpr int BK = BR * FK;      // 1 KB fragments per batch
  static constexpr int PF = BK >= 6 ? 4 : 2;  // fragments read ahead of the MFMAs
  static constexpr int RING_BYTES = D * BK * 1024;
  static constexpr int META_BYTES = MD * BR * 128;
  static_assert((MD & (MD - 1)) == 0, "index ring: power of two");
  static_assert((D - 3) * (1 + BK) <= 63, "vmcnt field");
  static_assert(NC == 1 || NC == 2, "a compute wave owns one or two 16-column tiles");
};

template <int NCTT, int NC, int KC>
__global__ void __launch_bounds__(64 * (NCTT / NC + 1), 1) conv_ring6f_kernel(ConvArgs a) {
  using C = RingFCfg<NCTT, NC, KC>;
  constexpr int BR = C::BR, D = C::D, MD = C::MD, FK = C::FK, BK = C::BK, PF = C::PF;
  const int T = a.tile_rows;
  PCS_RT(const long long tr_entry = __builtin_readcyclecounter(); long long tr_start = 0, tr_end = 0, tr_bar = 0, tr_wait = 0, tr_issue = 0; int tr_n = 0;)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // [A ring | pair-index ring | offset lists | batch descriptors | accumulator tile]: the DMA targets come first (low LDS addresses)
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
  const int NS = a.cin / 32, NCH = NS / KC;

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
  for (int e = 0; e < nk; ++e) {  // batch descriptors: job-major ((offset, chunk) outer, row blocks inner)
    const int m = kl_m[e], s0 = kl_s[e], nrb = (m + 15) >> 4, nbe = (nrb + BR - 1) / BR, base = kl_b[e];
    for (int idx = tid; idx < NCH * nbe; idx += C::NT) {
      const int c = idx / nbe, bb = idx - c * nbe, left = m - bb * BR * 16;
      bt[base + idx] = make_int2(s0 + bb * BR * 16, (left < 255 ? left : 255) | (c << 8));
    }
  }
  __syncthreads();

  if (NB > 0) {
    if (wid == C::NWC) {
      // ---------------- loader (as conv_ring6h.hip): pair indices and gathered rows by LDS-DMA, D - 2 batches ahead ----------------
      constexpr int LA = D - 2, LP = 2 * D - 3, W = D - 3;
      static_assert(MD >= LP + 1, "index ring too short");
      const int prow = (lane >> 5) * 16 + l15;                           // index DMA: lanes 0-31 row block 0, 32-63 row block 1
      const int pcol = ((lane >> 4) & 1) ? 1 - a.src_col : a.src_col;     // 16 src rows then 16 dst rows per row block
      const int32_t *pbase = a.pairs + pcol;
      const char *abase = reinterpret_cast<const char *>(a.src) + 16 * g;
      const int64_t row_bytes = (int64_t)a.cin * 4;
      auto desc = [&](int x) { return bt[x < NB ? x : NB - 1]; };        // same value in every lane
      auto idx_of = [&](int x, int r) { return meta[(x & (MD - 1)) * (BR * 32) + r * 32 + l15]; };
      auto issue_P = [&](int2 d, int x) {  // one DMA: the (src, dst) rows of the batch's 2 x 16 pairs; padding rows re-read the slice's last pair
        const int first = __builtin_amdgcn_readfirstlane(d.x), left = __builtin_amdgcn_readfirstlane(d.y) & 255;
        const int rk = prow < left ? prow : left - 1;
        glds4(pbase + (int64_t)(first + rk) * 2, __builtin_amdgcn_readfirstlane(meta_lds + (unsigned)(x & (MD - 1)) * (BR * 128)));
      };
      auto issue_A = [&](int2 d, const int (&idx)[BR], int slot) {  // BR * FK DMAs of 1 KB: fragment (r, q) = 16 rows x channels 32 c KC + 16 q + 4 g .. + 3
        const int c = __builtin_amdgcn_readfirstlane(d.y) >> 8;
        const unsigned dst = __builtin_amdgcn_readfirstlane(ring_lds + (unsigned)(slot * BK * 1024));
#pragma unroll
        for (int r = 0; r < BR; ++r) {
          const char *base = abase + (int64_t)idx[r] * row_bytes + c * (KC * 128);
#pragma unroll
          for (int q = 0; q < FK; ++q) glds16(base + q * 64, dst + (unsigned)((r * FK + q) * 1024));
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
      PCS_RT(tr_start = __builtin_readcyclecounter();)
      for (int i = -LA; i < NB; ++i) {
        PCS_RT(const long long t0 = __builtin_readcyclecounter();)
        if (i >= 0) ring_barrier();
        PCS_RT(const long long t1 = __builtin_readcyclecounter(); tr_bar += t1 - t0; ++tr_n;)
        const int2 dA_n = desc(i + 1 + LA), dP_n = desc(i + 1 + LP);  // next iteration's operands, read behind this one's DMA issue
        int idx_n[BR];
#pragma unroll
        for (int r = 0; r < BR; ++r) idx_n[r] = idx_of(i + 1 + LA, r);
        if (i + LA < NB) {
          issue_P(dP, i + LP);
          if (!(PCS_RINGF_ABLATE & 32)) issue_A(dA, idx, slot);
          slot = slot + 1 == D ? 0 : slot + 1;
          PCS_RT(const long long t2 = __builtin_readcyclecounter(); tr_issue += t2 - t1;)
          asm volatile("s_waitcnt vmcnt(%0)" ::"i"(W * (1 + BK)) : "memory");
          PCS_RT(tr_wait += __builtin_readcyclecounter() - t2;)
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        dA = dA_n; dP = dP_n;
#pragma unroll
        for (int r = 0; r < BR; ++r) idx[r] = idx_n[r];
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // nothing of this workgroup's DMA may land after its LDS is released
      ring_barrier();  // closing barrier: the compute waves' last batch passes the same barrier every batch does
      PCS_RT(tr_end = __builtin_readcyclecounter();)
    } else {
      // ---------------- compute wave: NC 16-column tiles, every row block of every offset ------------------------------------
      const int col0 = h_local_col(NCTT, wid * NC, l15);   // NC = 2: the wave's tile pair owns columns (col0, col0 + 1)
      const unsigned col_off = 4u * (unsigned)col0;
      const int NJ = nk * NCH;
      typedef std::conditional_t<NC == 2, v2f, float> bvec;
      bvec B[2][FK][4];   // weights of the running job / the next one: channel 16 q + 4 g + e of the chunk, this lane's column(s)
      const float *wcol = a.W + n0 + col0;
      int nloaded = 0;
      auto load_B = [&](auto set_tag, int e, int c) {
        constexpr int S = decltype(set_tag)::value;
        if ((PCS_RINGF_ABLATE & 16) && nloaded >= 2) return;
        ++nloaded;
        const int k = __builtin_amdgcn_readfirstlane(kl_k[e]);
        const float *wk = wcol + ((int64_t)k * a.cin + c * (KC * 32) + 4 * g) * a.cout;
#pragma unroll
        for (int q = 0; q < FK; ++q)
#pragma unroll
          for (int e4 = 0; e4 < 4; ++e4)
            B[S][q][e4] = *reinterpret_cast<const bvec *>(wk + (int64_t)(16 * q + e4) * a.cout);
      };
      auto read_frag = [&](int slot, int f) {
        return *(PCS_LDS(const v4f) *)(size_t)(ring_lds + (unsigned)((slot * BK + f) * 1024 + lane * 16));
      };
      // commit of this wave's columns of 16 dst rows in two parts, so that the LDS round trip hides behind MFMAs: the reads are
      // issued one fragment (eight MFMAs) before the adds and writes that consume them
      typedef std::conditional_t<NC == 2, v2f, float> cvec;
      auto commit_read = [&](cvec (&v)[4], const unsigned (&rw)[4]) {
        if (PCS_RINGF_ABLATE & (1 | 64)) { for (int j = 0; j < 4; ++j) v[j] = cvec{}; return; }
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = *(PCS_LDS(const cvec) *)(size_t)rw[j];
      };
      auto commit_finish = [&](cvec (&v)[4], const f32x4 (&ac)[NC], const unsigned (&rw)[4]) {
        if (PCS_RINGF_ABLATE & 1) { if (ac[0][0] == 1234.5f) *(PCS_LDS(float) *)(size_t)rw[0] = ac[0][1]; return; }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if constexpr (NC == 2) { v[j].x += ac[0][j]; v[j].y += ac[1][j]; }
          else v[j] += ac[0][j];
        }
        if (PCS_RINGF_ABLATE & 128) { if (ac[0][0] == 1234.5f) *(PCS_LDS(float) *)(size_t)rw[0] = ((const float *)&v[1])[0] + ((const float *)&v[2])[0] + ((const float *)&v[3])[0]; return; }
        if (PCS_RINGF_ABLATE & 256) {  // writes to a fixed conflict-free place of this wave (debug)
#pragma unroll
          for (int j = 0; j < 4; ++j) *(PCS_LDS(cvec) *)(size_t)(acc_lds + (unsigned)((wid * 4 + j) * 512 + lane * 8)) = v[j];
          return;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) *(PCS_LDS(cvec) *)(size_t)rw[j] = v[j];
      };
      auto row_addrs = [&](unsigned (&rw)[4], const v4i &d4, int rb, int m) {  // LDS byte addresses of the four dst rows this lane commits
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int rk = rb * 16 + 4 * g + j;
          const int dl = rk < m ? d4[j] - (int)row0 : T;   // padding rows (and the row block a partial batch lacks) go to the sink row
          rw[j] = acc_lds + __umul24((unsigned)dl, (unsigned)(C::ACS * 4)) + col_off;
        }
      };
      auto read_dst = [&](v4i (&d4)[BR], int x) {
#pragma unroll
        for (int r = 0; r < BR; ++r)
          d4[r] = *(PCS_LDS(const v4i) *)(size_t)(meta_lds + (unsigned)((x & (MD - 1)) * (BR * 128) + r * 128 + 64 + g * 16));
      };

      int i = 0, slot = 0;               // batch, its ring slot
      int je = 0, jc = 0;                // (offset, chunk) of the running job
      int pe = 0, pc = 0, pj = 0;        // job whose weights are loaded next (clamped to the last one)
      auto advance = [&](int &e, int &c) { if (++c == NCH) { c = 0; ++e; } };
      v4f afr[BK];
      f32x4 acc1p[NC];                   // row block 1 of the previous batch: committed behind the first MFMAs of the next
      unsigned rows1p[4];
#pragma unroll
      for (int tt = 0; tt < NC; ++tt) acc1p[tt] = (f32x4){0, 0, 0, 0};
#pragma unroll
      for (int j = 0; j < 4; ++j) rows1p[j] = acc_lds + (unsigned)(T * C::ACS * 4) + col_off;   // the first batch has no predecessor: zeros into the sink row

      // weight loads are unconditional (past the last job they re-read its slab): the compiler's vmcnt for the running job's
      // operands then counts the same number of younger loads on every path (conv_ring6h.hip)
      auto next_job = [&]() { if (pj + 1 < NJ) advance(pe, pc); ++pj; };
      load_B(std::integral_constant<int, 0>{}, pe, pc); next_job();
      ring_barrier();  // barrier 0: batch 0 has landed
      PCS_RT(tr_start = __builtin_readcyclecounter();)
#pragma unroll
      for (int f = 0; f < PF; ++f) afr[f] = read_frag(0, f);
      v4i dnext[BR];                     // dst rows of the batch about to run (read one batch ahead, behind the barrier that publishes them)
      read_dst(dnext, 0);

      // One batch, straight-line: BK fragments of 4 NC MFMAs each. Everything else a batch needs -- the ring reads PF fragments
      // ahead, the dst rows of the next batch, row addresses, the two commits (LDS reads one fragment ahead of their adds and
      // writes; row block 1's commit rides in the next batch) -- is attached to a fragment and INTERLEAVED with that
      // fragment's MFMAs by an explicit pipeline (sched_group_barrier: one MFMA, then up to VPM VALU and one LDS operation):
      // a wave issues in order, so side work left in a cluster between MFMA groups runs beside the last MFMA only -- the
      // first version of this kernel spent 2050 cycles of a 3770-cycle batch there (profiles/round4_ring.md).
      constexpr int NMF = 4 * NC, VPM = NC == 2 ? 3 : 5;
      auto body = [&](auto set_tag, int rb0, int m) {
        constexpr int S = decltype(set_tag)::value;
        const int nslot = slot + 1 == D ? 0 : slot + 1;
        unsigned rows0[4], rows1[4];
        f32x4 acc[BR][NC];
        cvec cv[4];
#pragma unroll
        for (int r = 0; r < BR; ++r)
#pragma unroll
          for (int tt = 0; tt < NC; ++tt) acc[r][tt] = (f32x4){0, 0, 0, 0};
#pragma unroll
        for (int f = 0; f < BK; ++f) {
          const int r = f / FK, q = f % FK;
          if (f == BK - PF) {
            PCS_RT(const long long tb0 = __builtin_readcyclecounter();)
            ring_barrier();  // barrier i + 1: the next batch has landed (after the last batch: the loader's closing barrier)
            PCS_RT(tr_bar += __builtin_readcyclecounter() - tb0; ++tr_n;)
            read_dst(dnext, i + 1);   // rows0 / rows1 of this batch were formed at fragments 0 / 1
          }
          if (!(PCS_RINGF_ABLATE & 2)) {
            if (f + PF < BK) afr[f + PF] = read_frag(slot, f + PF);
            else afr[f + PF - BK] = read_frag(nslot, f + PF - BK);   // behind the last batch: a slot nobody reads again
          }
          if (f == 0) { commit_read(cv, rows1p); row_addrs(rows0, dnext[0], rb0, m); }
          if (f == 1) { commit_finish(cv, acc1p, rows1p); row_addrs(rows1, dnext[1], rb0 + 1, m); }
          if (f == FK) commit_read(cv, rows0);
          if (f == FK + 1) commit_finish(cv, acc[0], rows0);
#pragma unroll
          for (int e4 = 0; e4 < 4; ++e4) {
            const float av = afr[f][e4];
            if (PCS_RINGF_ABLATE & 4) { acc[r][0][0] += av * ((const float *)&B[S][q][e4])[0]; continue; }
            if constexpr (NC == 2) {
              acc[r][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, B[S][q][e4].x, acc[r][0], 0, 0, 0);
              acc[r][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, B[S][q][e4].y, acc[r][1], 0, 0, 0);
            } else {
              acc[r][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, B[S][q][e4], acc[r][0], 0, 0, 0);
            }
          }
          if (!(PCS_RINGF_ABLATE & 512)) {
#pragma unroll
            for (int mf = 0; mf < NMF; ++mf) {
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // one MFMA
              __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);   // VALU
              __builtin_amdgcn_sched_group_barrier(0x080, 1, 0);     // one LDS operation
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int tt = 0; tt < NC; ++tt) acc1p[tt] = acc[1][tt];
#pragma unroll
        for (int j = 0; j < 4; ++j) rows1p[j] = rows1[j];
        ++i;
        slot = nslot;
      };
      auto run_job = [&](auto set_tag) {  // one (offset, chunk): its weights are in register set S; load the next job's into the other set
        constexpr int S = decltype(set_tag)::value;
        load_B(std::integral_constant<int, 1 - S>{}, pe, pc); next_job();
        const int m = __builtin_amdgcn_readfirstlane(kl_m[je]);
        const int nrb = (m + 15) >> 4;
        int rb0 = 0;  // m > 0: at least one batch (a do-while: set S is provably consumed before it is loaded again)
        do { body(set_tag, rb0, m); rb0 += BR; } while (rb0 < nrb);
        advance(je, jc);
      };
      for (int jg = NJ / 2; jg > 0; --jg) {  // jobs in pairs (the two register sets alternate statically), one loop exit
        run_job(std::integral_constant<int, 0>{});
        run_job(std::integral_constant<int, 1>{});
      }
      if (NJ % 2) run_job(std::integral_constant<int, 0>{});
      { cvec cv[4]; commit_read(cv, rows1p); commit_finish(cv, acc1p, rows1p); }
      PCS_RT(tr_end = __builtin_readcyclecounter();)
    }
  }
  __syncthreads();
  // epilogue: fp32 tile (+ bias), 16-byte stores, every dst row written once (shared with conv_wave5.hip)
  const int rows = (int)((a.n_dst - row0) < (int64_t)T ? (a.n_dst - row0) : (int64_t)T);
  float *drow = a.dst + row0 * a.cout + n0;
  const int ldd = a.cout;
  conv_tile_epilogue<C::CT, C::NT>(acc_l, C::ACS, rows, n0, a.cout, a.bias, a.stats ? a.stats + tile * 2 * a.cout : nullptr, tid,
                                   [&](int r, int cq, const float4 &v) {
                                     *reinterpret_cast<float4 *>(drow + (int64_t)r * ldd + cq) = v;
                                     return v;
                                   });
#if PCS_RING_TRACE
  if (lane == 0 && blockIdx.x < kRingTraceBlocks && g_ringf_trace) {
    long long *t = g_ringf_trace + ((int64_t)blockIdx.x * 8 + wid) * 8;
    t[0] = tr_entry; t[1] = tr_start; t[2] = tr_end; t[3] = __builtin_readcyclecounter(); t[4] = tr_bar; t[5] = tr_wait; t[6] = tr_issue; t[7] = tr_n;
  }
#endif
}

#if PCS_RING_TRACE
long long *g_ringf_trace_host = nullptr;
void ringf_trace_prepare(hipStream_t st) {
  if (!g_ringf_trace_host) {
    (void)hipMalloc(&g_ringf_trace_host, (size_t)kRingTraceBlocks * 64 * sizeof(long long));
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_ringf_trace), &g_ringf_trace_host, sizeof(g_ringf_trace_host));
  }
  (void)hipMemsetAsync(g_ringf_trace_host, 0, (size_t)kRingTraceBlocks * 64 * sizeof(long long), st);
}
#endif

template <int NCTT, int NC, int KC>
int launch_ringf(const ConvArgs &a, size_t lds, hipStream_t st) {
  using C = RingFCfg<NCTT, NC, KC>;
  const int64_t nblocks = a.order ? ceil_div(a.ntiles, 8) * 8 * a.ncoltiles : a.ntiles * a.ncoltiles;
  if (nblocks <= 0) return PCS_OK;
  if (nblocks > 0x7FFFFFFF) { set_error("pcs_conv_gather_gemm_f32: grid too large"); return PCS_EUNSUPPORTED; }
  auto kern = conv_ring6f_kernel<NCTT, NC, KC>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxDynLds);
    attr_set = true;
  }
  PCS_RT(ringf_trace_prepare(st);)
  hipLaunchKernelGGL(kern, dim3((unsigned)nblocks), dim3(C::NT), lds, st, a);
  return check_launch("pcs_conv_gather_gemm_f32(ring6f)");
}

}  // namespace

#if PCS_RING_TRACE
// variant builds only: phase timers of the last fp32 ring launch, [block][wave][8] (see g_ringf_trace)
extern "C" int pcs_debug_ringf_trace(long long *host_out) {
  if (!g_ringf_trace_host || !host_out) return PCS_EINVAL;
  if (hipDeviceSynchronize() != hipSuccess) return PCS_ELAUNCH;
  return hipMemcpy(host_out, g_ringf_trace_host, (size_t)kRingTraceBlocks * 64 * sizeof(long long), hipMemcpyDeviceToHost) == hipSuccess ? PCS_OK : PCS_ELAUNCH;
}
#endif

namespace pcs {

int &conv_ringf_mode() {
  static int mode = getenv("PCS_CONV_RINGF") ? atoi(getenv("PCS_CONV_RINGF")) : -1;
  return mode;
}

int launch_conv_ring6f(const ConvArgs &a0, hipStream_t st) {
  RingShape s;
  if (!conv_ringf_applies(a0.cin, a0.cout, a0.K, a0.tile_rows, &s)) { set_error("pcs_conv_gather_gemm_f32(ring6f): shape not served"); return PCS_EUNSUPPORTED; }
  ConvArgs a = a0;
  a.ncoltiles = (int)ceil_div(a.cout, 16 * s.nctt);
  a.xcd_remap = 1;
  int acc_off = 0;
  const size_t lds = conv_ringf_lds(a.tile_rows, s, a.cin / 32, a.K, &acc_off);
  a.ring_acc_off = acc_off;
#define PCS_RINGF(N, C_)                                          \
  if (s.nctt == N && s.nc == C_) return s.kc == 2 ? launch_ringf<N, C_, 2>(a, lds, st) : launch_ringf<N, C_, 1>(a, lds, st);
  PCS_RINGF(4, 1)
  PCS_RINGF(6, 2)
  PCS_RINGF(8, 2)
#undef PCS_RINGF
  set_error("pcs_conv_gather_gemm_f32(ring6f): unreachable");
  return PCS_EINVAL;
}

}  // namespace pcs
