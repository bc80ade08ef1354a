// spdownsample key packing + rulebook (kernel map) builder -- gfx950, integer, bit-exact.
// Reference semantics: TS:torchsparse/nn/functional/downsample.py:11-52 and
// TS:torchsparse/nn/functional/conv.py:156-176 (kernel_hash -> hashquery -> sum/nonzero).
// The (K,N) int64 hash matrix and the (K,N) int64 result matrix of the reference are never
// materialised: the offset hash is formed in registers and probed at once; the only (K,N)
// intermediate is an int32 result matrix that pass 2 compacts in the reference's order
// (k-major, query row ascending) with wave ballots + a block-count prefix scan.
#include "pcs_common.h"

using namespace pcs;

namespace {

constexpr int kRowsPerBlock = 1024;  // 256 threads x 4 rounds
constexpr int64_t kBias = 1 << 17;

// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int64_t pack_key(int x, int y, int z, int b, int32_t *err) {
  const int64_t bx = (int64_t)x + kBias, by = (int64_t)y + kBias, bz = (int64_t)z + kBias;
  if (bx < 0 || bx >= 2 * kBias || by < 0 || by >= 2 * kBias || bz < 0 || bz >= 2 * kBias ||
      b < 0 || b > 511) {
    *err = 1;
    return INT64_MAX;
  }
  return ((int64_t)b << 54) | (bx << 36) | (by << 18) | bz;
}

__global__ void __launch_bounds__(256) ds_pack_fast_kernel(const int4 *__restrict__ coords,
                                                           int64_t n, int sx, int sy, int sz,
                                                           int64_t *__restrict__ keys,
                                                           int32_t *err) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    int4 c = coords[i];
    // downsample.py:25-28: div(...).trunc() * stride  == C integer division (toward zero)
    keys[i] = pack_key((c.x / sx) * sx, (c.y / sy) * sy, (c.z / sz) * sz, c.w, err);
  }
}

// general branch (downsample.py:29-45): candidates coords + offsets[k]; blockIdx.y = k
__global__ void __launch_bounds__(256) ds_pack_general_kernel(
    const int4 *__restrict__ coords, int64_t n, int sx, int sy, int sz,
    const int32_t *__restrict__ offsets, const int32_t *__restrict__ cmin, int K,
    int64_t *__restrict__ keys, int32_t *err) {
  const int k = blockIdx.y;
  const int ox = offsets[3 * k], oy = offsets[3 * k + 1], oz = offsets[3 * k + 2];
  const int mx = cmin[0], my = cmin[1], mz = cmin[2];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    int4 c = coords[i];
    const int x = c.x + ox, y = c.y + oy, z = c.z + oz;
    const bool keep = (x % sx == 0) && (y % sy == 0) && (z % sz == 0) && x >= mx && y >= my && z >= mz;
    // reference layout of the candidate list is row-major (i, k); order is irrelevant (unique sorts)
    keys[i * K + k] = keep ? pack_key(x, y, z, c.w, err) : INT64_MAX;
  }
}

__global__ void __launch_bounds__(256) ds_unpack_kernel(const int64_t *__restrict__ keys,
                                                        int64_t m, int4 *__restrict__ coords) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t k = keys[i];
    int4 c;
    c.w = (int)(k >> 54);
    c.x = (int)(((k >> 36) & (2 * kBias - 1)) - kBias);
    c.y = (int)(((k >> 18) & (2 * kBias - 1)) - kBias);
    c.z = (int)((k & (2 * kBias - 1)) - kBias);
    coords[i] = c;
  }
}

// ---------------------------------------------------------------------------------------------
// pass 1: probe. grid (nblk, K). Each block covers kRowsPerBlock consecutive query rows.
// SYM: submanifold map (query coordinates == table coordinates, offsets[K-1-k] == -offsets[k], K odd). A hit
// (row j + offsets[k] -> row i) IS the hit (row i + offsets[K-1-k] -> row j), and the centre offset is the identity: only
// the first K/2 offsets are probed (grid.y = K/2 + 1), the mirrored half of the hit matrix is scattered (its rows
// pre-filled with -1) and its block counts are taken by a streaming pass over it (rb_count_kernel) -- half the random
// table probes, which are what this kernel costs (1.9 ms per training step at 2 TB/s of sectors).
template <bool SYM>
__global__ void __launch_bounds__(256) rb_probe_kernel(const int4 *__restrict__ q, int64_t nq,
                                                       const int32_t *__restrict__ offsets,
                                                       TableView tab,
                                                       int32_t *__restrict__ results,
                                                       int32_t *__restrict__ blockcnt, int K) {
  const int k = blockIdx.y;
  const int ox = offsets[3 * k], oy = offsets[3 * k + 1], oz = offsets[3 * k + 2];
  const int64_t base = (int64_t)blockIdx.x * kRowsPerBlock;
  int32_t *res = results + (int64_t)k * nq;
  const bool centre = SYM && k == K / 2;
  int32_t *mres = results + (int64_t)(K - 1 - k) * nq;           // SYM: row K-1-k of the hit matrix
  int hits = 0;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int64_t j = base + r * 256 + threadIdx.x;
    int32_t v = -1;
    if (j < nq) {
      if (centre) {
        v = (int32_t)j;
      } else {
        int4 c = q[j];
        v = table_lookup(tab, fnv60(c.x + ox, c.y + oy, c.z + oz, c.w));
      }
      res[j] = v;
      if (SYM && !centre && v >= 0) mres[v] = (int32_t)j;
    }
    hits += (v >= 0);
  }
  // block reduction of hit counts
  __shared__ int wsum[4];
  for (int o = 32; o > 0; o >>= 1) hits += __shfl_down(hits, o, 64);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = hits;
  __syncthreads();
  if (threadIdx.x == 0)
    blockcnt[(int64_t)k * gridDim.x + blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// block counts of rows k0 .. k0 + gridDim.y - 1 of the hit matrix (the scattered half of a symmetric probe)
__global__ void __launch_bounds__(256) rb_count_kernel(const int32_t *__restrict__ results, int64_t nq, int k0,
                                                       int32_t *__restrict__ blockcnt) {
  const int k = k0 + blockIdx.y;
  const int64_t base = (int64_t)blockIdx.x * kRowsPerBlock;
  const int32_t *res = results + (int64_t)k * nq;
  int hits = 0;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int64_t j = base + r * 256 + threadIdx.x;
    hits += (j < nq && res[j] >= 0);
  }
  __shared__ int wsum[4];
  for (int o = 32; o > 0; o >>= 1) hits += __shfl_down(hits, o, 64);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = hits;
  __syncthreads();
  if (threadIdx.x == 0) blockcnt[(int64_t)k * gridDim.x + blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// pass 1b: single-workgroup exclusive scan of the K*nblk block counts (<= ~30k entries),
// then nbsizes[k] / koff[k].
__global__ void __launch_bounds__(1024) rb_scan_kernel(const int32_t *__restrict__ cnt,
                                                       int32_t *__restrict__ off, int64_t total,
                                                       int nblk, int K,
                                                       int64_t *__restrict__ nbsizes,
                                                       int32_t *__restrict__ koff) {
  __shared__ int wtot[16];
  __shared__ int carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  for (int64_t c0 = 0; c0 < total; c0 += 1024) {
    const int64_t i = c0 + threadIdx.x;
    const int v = (i < total) ? cnt[i] : 0;
    int s = v;  // inclusive wave scan
    for (int o = 1; o < 64; o <<= 1) {
      int t = __shfl_up(s, o, 64);
      if (lane >= o) s += t;
    }
    if (lane == 63) wtot[wid] = s;
    __syncthreads();
    if (wid == 0) {
      int t = (lane < 16) ? wtot[lane] : 0;
      int ts = t;
      for (int o = 1; o < 16; o <<= 1) {
        int u = __shfl_up(ts, o, 64);
        if (lane >= o) ts += u;
      }
      if (lane < 16) wtot[lane] = ts - t;  // exclusive wave offsets
    }
    __syncthreads();
    const int carry = carry_s;
    const int excl = carry + wtot[wid] + s - v;
    if (i < total) off[i] = excl;
    __syncthreads();
    if (threadIdx.x == 1023) carry_s = excl + v;
    __syncthreads();
  }
  if (threadIdx.x == 0) off[total] = carry_s;
  __syncthreads();
  for (int k = threadIdx.x; k <= K; k += 1024) {
    const int32_t a = off[(int64_t)k * nblk];  // k == K -> off[total]
    koff[k] = a;
    if (k < K) nbsizes[k] = (int64_t)off[(int64_t)(k + 1) * nblk] - a;
  }
}

// pass 2: ordered compaction. grid (nblk, K).
__global__ void __launch_bounds__(256) rb_fill_kernel(const int32_t *__restrict__ results,
                                                      int64_t nq,
                                                      const int32_t *__restrict__ blockoff,
                                                      int32_t *__restrict__ pairs) {
  const int k = blockIdx.y;
  const int64_t base = (int64_t)blockIdx.x * kRowsPerBlock;
  const int32_t *res = results + (int64_t)k * nq;
  __shared__ int wcnt[4];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  int64_t out = blockoff[(int64_t)k * gridDim.x + blockIdx.x];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int64_t j = base + r * 256 + threadIdx.x;
    const int32_t v = (j < nq) ? res[j] : -1;
    const unsigned long long m = __ballot(v >= 0);
    const int rank = __popcll(m & ((1ULL << lane) - 1ULL));
    if (lane == 0) wcnt[wid] = __popcll(m);
    __syncthreads();
    int woff = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const int c = wcnt[w];
      if (w < wid) woff += c;
      tot += c;
    }
    if (v >= 0) {
      const int64_t p = out + woff + rank;
      reinterpret_cast<int2 *>(pairs)[p] = make_int2(v, (int)j);
    }
    out += tot;
    __syncthreads();
  }
}

// seg[k*(nt+1) + t] = first pair index of offset k whose dst row >= t*tile_rows
__global__ void __launch_bounds__(256) rb_segments_kernel(const int32_t *__restrict__ pairs,
                                                          const int32_t *__restrict__ koff,
                                                          int K, int64_t nt1, int tile_rows,
                                                          int dst_col, int32_t *__restrict__ seg) {
  const int64_t total = (int64_t)K * nt1;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int k = (int)(e / nt1);
    const int64_t t = e - (int64_t)k * nt1;
    const int64_t target = t * tile_rows;
    int lo = koff[k], hi = koff[k + 1];
    while (lo < hi) {
      const int mid = lo + ((hi - lo) >> 1);
      if ((int64_t)pairs[2 * (int64_t)mid + dst_col] < target) lo = mid + 1; else hi = mid;
    }
    seg[e] = lo;
  }
}

// Launch order of the row tiles of one segment table: heaviest first, work(t) = 16-row blocks of tile t over all offsets (what
// the fused convolution issues MFMAs for). One workgroup: per-tile work (coalesced over the tiles of one offset, summed
// with LDS atomics), then a counting sort by work (histogram, descending prefix, scatter). The order among equally heavy
// tiles is arbitrary -- it only decides which workgroup slot runs which tile, never a sum.
// xcd != 0 [r6, opt-in: PCS_TILE_ORDER_XCD=1]: the conv kernels run launch position p on XCD p % 8 (conv_wave5.hip: slot mapping).
// Every XCD gets one CONTIGUOUS eighth of the tiles (lengths differ by at most one, so position j * 8 + c is the j-th tile of
// eighth c) and walks it heaviest first: the tiles in flight on an XCD still come from one stretch of rows and share its L2.
// Measured (profiles/round6_conv_xcd_order_ab2.txt, round6_conv_order_traffic2.txt, round6_order_bench_ab.txt): per launch the
// speed of the chip-wide order (+4-8 % over row order on the dense levels) with 11-33 % less HBM read traffic; over a whole
// training step 0.6 % (fp32) / 0.9 % (bf16) SLOWER than the chip-wide order, twice out of twice, and the step's conv traffic only
// 1-3 % lower (the full-resolution levels, in row order either way, carry most of it) -> not the default.
constexpr int kOrderBins = 2048;    // work <= K * (tile_rows / 16 + 1) <= 32 * 33
constexpr int kOrderTiles = 12288;  // tiles whose work fits the LDS (48 KB); more: the work is recomputed per pass
__global__ void __launch_bounds__(1024) rb_tile_order_kernel(const int32_t *__restrict__ seg, int K, int64_t ntiles,
                                                             int32_t *__restrict__ order, int xcd) {
  __shared__ int hist[kOrderBins];
  __shared__ int base[kOrderBins];
  __shared__ int work_l[kOrderTiles];
  __shared__ int wsum[16];
  const int tid = threadIdx.x;
  const int64_t nt1 = ntiles + 1;
  const bool in_lds = ntiles <= kOrderTiles;
  auto clampw = [](int w) { return w < kOrderBins - 1 ? w : kOrderBins - 1; };
  auto work_of = [&](int64_t t) {  // the slow way (one thread walks the offsets of a tile)
    int w = 0;
    for (int k = 0; k < K; ++k) {
      const int64_t o = (int64_t)k * nt1 + t;
      w += (seg[o + 1] - seg[o] + 15) >> 4;
    }
    return clampw(w);
  };
  if (in_lds) {
    for (int i = tid; i < (int)ntiles; i += 1024) work_l[i] = 0;
    __syncthreads();
    const int64_t total = (int64_t)K * ntiles;
    for (int64_t e = tid; e < total; e += 1024) {  // consecutive threads: consecutive tiles of one offset
      const int k = (int)(e / ntiles);
      const int t = (int)(e - (int64_t)k * ntiles);
      const int64_t o = (int64_t)k * nt1 + t;
      const int nb = (seg[o + 1] - seg[o] + 15) >> 4;
      if (nb) atomicAdd(&work_l[t], nb);
    }
  }
  const int nparts = xcd ? 8 : 1;
  const int64_t q = ntiles / nparts, r = ntiles % nparts;
  for (int c = 0; c < nparts; ++c) {  // one counting sort per part: tiles [lo, lo + ln) -> positions j * nparts + c
    const int64_t lo = c * q + (c < r ? c : r), ln = q + (c < r ? 1 : 0);
    __syncthreads();
    for (int i = tid; i < kOrderBins; i += 1024) hist[i] = 0;
    __syncthreads();
    for (int64_t t = lo + tid; t < lo + ln; t += 1024) atomicAdd(&hist[in_lds ? clampw(work_l[t]) : work_of(t)], 1);
    __syncthreads();
    // base[w] = number of tiles heavier than w: two bins per thread, scanned from the heavy end
    {
      const int b0 = kOrderBins - 1 - 2 * tid, b1 = b0 - 1;  // this thread's bins, heavy first
      const int h0 = hist[b0], h1 = hist[b1];
      int incl = h0 + h1;
      const int lane = tid & 63, wid = tid >> 6;
      for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o, 64);
        if (lane >= o) incl += v;
      }
      if (lane == 63) wsum[wid] = incl;
      __syncthreads();
      int off = 0;
      for (int w = 0; w < wid; ++w) off += wsum[w];
      const int excl = off + incl - (h0 + h1);
      base[b0] = excl;
      base[b1] = excl + h0;
    }
    __syncthreads();
    for (int64_t t = lo + tid; t < lo + ln; t += 1024)
      order[(int64_t)atomicAdd(&base[in_lds ? clampw(work_l[t]) : work_of(t)], 1) * nparts + c] = (int32_t)t;
  }
}

struct RbWs {
  int32_t *blockcnt;  // K*nblk
  int32_t *blockoff;  // K*nblk + 1
  int32_t *koff;      // K + 1
  size_t bytes;
};

RbWs carve(void *ws, int64_t nq, int K) {
  const int64_t nblk = ceil_div(nq > 0 ? nq : 1, kRowsPerBlock);
  RbWs r;
  char *p = reinterpret_cast<char *>(ws);
  const size_t a = ((size_t)K * nblk * 4 + 255) & ~(size_t)255;
  const size_t b = (((size_t)K * nblk + 1) * 4 + 255) & ~(size_t)255;
  const size_t c = (((size_t)K + 1) * 4 + 255) & ~(size_t)255;
  r.blockcnt = reinterpret_cast<int32_t *>(p);
  r.blockoff = reinterpret_cast<int32_t *>(p + a);
  r.koff = reinterpret_cast<int32_t *>(p + a + b);
  r.bytes = a + b + c;
  return r;
}

}  // namespace

extern "C" int pcs_downsample_pack(const int32_t *coords, int64_t n, const int32_t *ss,
                                   int32_t mode, const int32_t *offsets, int32_t K,
                                   const int32_t *coords_min3, int64_t *keys, int32_t *err,
                                   void *stream) {
  // ss (sample_stride3) is a HOST pointer: three small ints known to the caller
  if (n < 0 || !ss || ss[0] <= 0 || ss[1] <= 0 || ss[2] <= 0 || !err) { set_error("pcs_downsample_pack: bad args"); return PCS_EINVAL; }
  if (n == 0) return PCS_OK;
  if (!coords || !keys || ((uintptr_t)coords & 15)) { set_error("pcs_downsample_pack: bad pointers"); return PCS_EINVAL; }
  hipStream_t st = as_stream(stream);
  if (mode == 0) {
    hipLaunchKernelGGL(ds_pack_fast_kernel, dim3(stream_grid(n, 256)), dim3(256), 0, st,
                       reinterpret_cast<const int4 *>(coords), n, ss[0], ss[1], ss[2], keys, err);
  } else {
    if (!offsets || !coords_min3 || K <= 0 || K > 65535) { set_error("pcs_downsample_pack: general branch needs offsets/min"); return PCS_EINVAL; }
    int gx = stream_grid(n, 256);
    if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(ds_pack_general_kernel, dim3(gx, K), dim3(256), 0, st,
                       reinterpret_cast<const int4 *>(coords), n, ss[0], ss[1], ss[2], offsets,
                       coords_min3, K, keys, err);
  }
  return check_launch("pcs_downsample_pack");
}

extern "C" int pcs_downsample_unpack(const int64_t *keys, int64_t m, int32_t *coords,
                                     void *stream) {
  if (m < 0) { set_error("pcs_downsample_unpack: bad size"); return PCS_EINVAL; }
  if (m == 0) return PCS_OK;
  if (!keys || !coords || ((uintptr_t)coords & 15)) { set_error("pcs_downsample_unpack: bad pointers"); return PCS_EINVAL; }
  hipLaunchKernelGGL(ds_unpack_kernel, dim3(stream_grid(m, 256)), dim3(256), 0, as_stream(stream),
                     keys, m, reinterpret_cast<int4 *>(coords));
  return check_launch("pcs_downsample_unpack");
}

extern "C" size_t pcs_rulebook_ws_bytes(int64_t nq, int32_t K) {
  return carve(nullptr, nq, K).bytes;
}

extern "C" int pcs_rulebook_probe(const int32_t *qcoords, int64_t nq, const int32_t *offsets,
                                  int32_t K, const void *table, int64_t capacity,
                                  int32_t *results, int64_t *nbsizes, void *ws, size_t ws_bytes,
                                  int32_t symmetric, void *stream) {
  if (nq < 0 || K <= 0 || K > 65535 || !table || capacity <= 0 || (capacity & (capacity - 1)) ||
      !offsets || !nbsizes || !ws) {
    set_error("pcs_rulebook_probe: bad args");
    return PCS_EINVAL;
  }
  if ((int64_t)K * (nq > 0 ? nq : 1) >= 0x7FFFFFFFLL) { set_error("pcs_rulebook_probe: K*nq exceeds int32 pair indexing"); return PCS_EUNSUPPORTED; }
  RbWs w = carve(ws, nq, K);
  if (ws_bytes < w.bytes) { set_error("pcs_rulebook_probe: workspace %zu < %zu", ws_bytes, w.bytes); return PCS_EWORKSPACE; }
  hipStream_t st = as_stream(stream);
  const int64_t nblk = ceil_div(nq > 0 ? nq : 1, kRowsPerBlock);
  if (nq > 0) {
    if (!qcoords || !results || ((uintptr_t)qcoords & 15)) { set_error("pcs_rulebook_probe: bad pointers"); return PCS_EINVAL; }
    if (symmetric && (K & 1) && K >= 3) {
      const int half = K / 2;  // rows half+1 .. K-1 are scattered by the probes of rows 0 .. half-1
      if (hipMemsetAsync(results + (int64_t)(half + 1) * nq, 0xFF, (size_t)half * nq * 4, st) != hipSuccess) {
        set_error("pcs_rulebook_probe: memset failed");
        return PCS_ELAUNCH;
      }
      hipLaunchKernelGGL(rb_probe_kernel<true>, dim3((unsigned)nblk, half + 1), dim3(256), 0, st,
                         reinterpret_cast<const int4 *>(qcoords), nq, offsets,
                         make_view(table, capacity), results, w.blockcnt, (int)K);
      hipLaunchKernelGGL(rb_count_kernel, dim3((unsigned)nblk, half), dim3(256), 0, st, results, nq, half + 1, w.blockcnt);
    } else {
      hipLaunchKernelGGL(rb_probe_kernel<false>, dim3((unsigned)nblk, K), dim3(256), 0, st,
                         reinterpret_cast<const int4 *>(qcoords), nq, offsets,
                         make_view(table, capacity), results, w.blockcnt, (int)K);
    }
  } else {
    if (hipMemsetAsync(w.blockcnt, 0, (size_t)K * nblk * 4, st) != hipSuccess) { set_error("pcs_rulebook_probe: memset failed"); return PCS_ELAUNCH; }
  }
  hipLaunchKernelGGL(rb_scan_kernel, dim3(1), dim3(1024), 0, st, w.blockcnt, w.blockoff,
                     (int64_t)K * nblk, (int)nblk, (int)K, nbsizes, w.koff);
  return check_launch("pcs_rulebook_probe");
}

extern "C" int pcs_rulebook_fill(const int32_t *results, int64_t nq, int32_t K, const void *ws,
                                 int32_t *pairs, int32_t *koff, void *stream) {
  if (nq < 0 || K <= 0 || !ws || !koff) { set_error("pcs_rulebook_fill: bad args"); return PCS_EINVAL; }
  RbWs w = carve(const_cast<void *>(ws), nq, K);
  hipStream_t st = as_stream(stream);
  if (hipMemcpyAsync(koff, w.koff, ((size_t)K + 1) * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) {
    set_error("pcs_rulebook_fill: koff copy failed");
    return PCS_ELAUNCH;
  }
  if (nq == 0) return PCS_OK;
  if (!results || !pairs) { set_error("pcs_rulebook_fill: null pointer"); return PCS_EINVAL; }
  const int64_t nblk = ceil_div(nq, kRowsPerBlock);
  hipLaunchKernelGGL(rb_fill_kernel, dim3((unsigned)nblk, K), dim3(256), 0, st, results, nq,
                     w.blockoff, pairs);
  return check_launch("pcs_rulebook_fill");
}

extern "C" int pcs_rulebook_tile_segments(const int32_t *pairs, const int32_t *koff, int32_t K,
                                          int64_t n_dst, int32_t tile_rows, int32_t dst_col,
                                          int32_t *seg, void *stream) {
  if (K <= 0 || n_dst < 0 || tile_rows <= 0 || (dst_col != 0 && dst_col != 1) || !koff || !seg) {
    set_error("pcs_rulebook_tile_segments: bad args");
    return PCS_EINVAL;
  }
  const int64_t nt1 = ceil_div(n_dst, tile_rows) + 1;
  hipLaunchKernelGGL(rb_segments_kernel, dim3(stream_grid((int64_t)K * nt1, 256)), dim3(256), 0,
                     as_stream(stream), pairs, koff, (int)K, nt1, (int)tile_rows, (int)dst_col, seg);
  return check_launch("pcs_rulebook_tile_segments");
}

extern "C" int pcs_rulebook_tile_order(const int32_t *seg, int32_t K, int64_t ntiles, int32_t *order, void *stream) {
  if (K <= 0 || K > 32 || ntiles < 0 || ntiles > 0x7FFFFFFF) { set_error("pcs_rulebook_tile_order: bad sizes"); return PCS_EINVAL; }
  if (ntiles == 0) return PCS_OK;
  if (!seg || !order) { set_error("pcs_rulebook_tile_order: null pointer"); return PCS_EINVAL; }
  // PCS_TILE_ORDER_XCD=1: heaviest first inside each XCD's contiguous eighth instead of chip-wide (see the kernel's comment:
  // equal per launch, 0.6-0.9 % slower over a training step, 11-33 % fewer HBM reads on the launches it applies to)
  static const int xcd = getenv("PCS_TILE_ORDER_XCD") ? atoi(getenv("PCS_TILE_ORDER_XCD")) : 0;
  hipLaunchKernelGGL(rb_tile_order_kernel, dim3(1), dim3(1024), 0, as_stream(stream), seg, (int)K, ntiles, order, xcd);
  return check_launch("pcs_rulebook_tile_order");
}
