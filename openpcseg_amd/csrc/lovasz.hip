// Lovasz-softmax (classes = 'present', per_image = False) with its gradient, all classes in ONE radix sort.
// Replaces the per-class python loop of R:tools/utils/common/lovasz_losses.py:158-204 (lovasz_softmax_flat: errors =
// |fg - p_c|, descending sort, lovasz_grad :23-35 of the sorted foreground, dot) that the reference's criterion runs on
// (N, num_class) probabilities every training step (R:pcseg/model/segmentor/voxel/minkunet/minkunet.py:336-356): there
// one stable sort, one gather, three scans and a dozen elementwise launches per class (19 x ~13 rocPRIM passes over
// 1.2 M keys at ~1 TB/s) and the same again through autograd; here
//   lovasz_count   valid points per class (integer: the 'present' test and gts of every class)
//   lovasz_keys    one 64-bit key per (class slot, point): slot << 32 | ~bits(error)  (errors are >= 0: their IEEE bits
//                  order like the values; the complement makes the ascending radix sort a DESCENDING one, ties in point
//                  order exactly as torch's stable descending sort leaves them); value = point | foreground << 31
//   rocprim::radix_sort_pairs  over all class slots at once, 32 + log2(slots) key bits (5 onesweep passes for <= 32 classes)
//   lovasz_blocksum / lovasz_final   the foreground cumsum per class (uniform segments: slot s owns [s n, (s + 1) n)), the
//                  Jaccard gradient from (gts, position, cumsum), the per-block loss partial in double and the gradient
//                  w.r.t. the probabilities scattered back by point into a class-major buffer: -+grad / n_present, zero where
//                  the error is zero (abs'(0) = 0: covers the ignored points, whose error is forced to zero so that they sort last)
//   lovasz_grad_rows  the class-major gradient turned into (n, num_class) rows
//   lovasz_reduce  mean over the present classes, deterministic order.
// HBM-bound integer / byte work (12 B per key-value, ~11 passes): no MFMA anywhere.
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "pcs_common.h"

namespace {
using namespace pcs;

constexpr int kMaxClasses = 60;   // (256 rows x (nc | 1) floats of the transpose tile must fit 64 KB)
constexpr int kRows = 256;        // points per workgroup in lovasz_keys
constexpr int kPerThread = 8;
constexpr int kBlockElems = 256 * kPerThread;   // sorted elements per workgroup in the scan kernels

struct Header {   // zeroed at the start of every call
  int32_t cnt[64];
};

struct Plan {
  int64_t n, total, nblk;
  int nc, ncp, skip, key_bits;
  size_t off_blocksum, off_blockloss, off_keys[2], off_vals[2], off_gradt, off_temp, temp_bytes, bytes;
};

size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }

int plan_for(int64_t n, int nc, int has_ignore, int64_t ignore, Plan &p) {
  if (n < 0 || nc < 1 || nc > kMaxClasses) {
    set_error("pcs_lovasz: %lld points x %d classes unsupported (1 <= classes <= %d)", (long long)n, nc, kMaxClasses);
    return PCS_EINVAL;
  }
  p.n = n; p.nc = nc;
  p.skip = (has_ignore && ignore >= 0 && ignore < nc) ? (int)ignore : -1;   // a class equal to the ignore label is never present
  p.ncp = nc - (p.skip >= 0 ? 1 : 0);
  p.total = n * p.ncp;
  if (n >= (1LL << 31) || p.total >= (1LL << 32) - 1) {
    set_error("pcs_lovasz: %lld points x %d classes exceed the 32-bit element index", (long long)n, p.ncp);
    return PCS_EINVAL;
  }
  p.key_bits = 32;
  while ((1 << (p.key_bits - 32)) < p.ncp) ++p.key_bits;
  p.nblk = ceil_div(n, kBlockElems);
  size_t off = align_up(sizeof(Header));
  p.off_blocksum = off; off = align_up(off + sizeof(int32_t) * p.nblk * p.ncp);
  p.off_blockloss = off; off = align_up(off + sizeof(double) * p.nblk * p.ncp);
  for (int i = 0; i < 2; ++i) { p.off_keys[i] = off; off = align_up(off + sizeof(uint64_t) * p.total); }
  for (int i = 0; i < 2; ++i) { p.off_vals[i] = off; off = align_up(off + sizeof(uint32_t) * p.total); }
  p.off_gradt = off; off = align_up(off + sizeof(float) * p.total);   // d loss / d probas, class-slot-major (see lovasz_final_kernel)
  p.off_temp = off;
  p.temp_bytes = 0;
  if (p.total > 0) {
    rocprim::double_buffer<uint64_t> k(nullptr, nullptr);
    rocprim::double_buffer<uint32_t> v(nullptr, nullptr);
    hipError_t e = rocprim::radix_sort_pairs(nullptr, p.temp_bytes, k, v, (size_t)p.total, 0u, (unsigned)p.key_bits, (hipStream_t)0);
    if (e != hipSuccess) {
      set_error("pcs_lovasz: rocprim temporary-storage query failed: %s", hipGetErrorString(e));
      return PCS_ELAUNCH;
    }
  }
  p.bytes = align_up(off + p.temp_bytes);
  return PCS_OK;
}

__device__ __forceinline__ bool label_valid(int64_t l, int nc, int has_ignore, int64_t ignore) {
  return l >= 0 && l < nc && !(has_ignore && l == ignore);
}

__global__ __launch_bounds__(256) void lovasz_count_kernel(const int64_t *__restrict__ labels, int64_t n, int nc, int has_ignore,
                                                          int64_t ignore, Header *hdr) {
  // lane c of every wave counts class c (nc <= 60 < 64) from the ballots of the wave's 64 labels: no same-address atomics
  __shared__ int32_t h[64];
  if (threadIdx.x < 64) h[threadIdx.x] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  int cnt = 0;
  const int64_t span = (int64_t)gridDim.x * 256;
  for (int64_t i0 = (int64_t)blockIdx.x * 256; i0 < n; i0 += span) {   // (uniform trip count over the wave)
    const int64_t i = i0 + threadIdx.x;
    const int64_t l = i < n ? labels[i] : -1;
    const int cls = label_valid(l, nc, has_ignore, ignore) ? (int)l : -1;
    for (int c = 0; c < nc; ++c) {
      const unsigned long long m = __ballot(cls == c);
      if (lane == c) cnt += __popcll(m);
    }
  }
  if (cnt) atomicAdd(&h[lane], cnt);
  __syncthreads();
  if (threadIdx.x < nc && h[threadIdx.x]) atomicAdd(&hdr->cnt[threadIdx.x], h[threadIdx.x]);
}

// one workgroup = 256 points: their probability rows come in coalesced, are turned through LDS (odd row stride: no bank
// conflicts) and leave as one key / value stream per class slot
__global__ __launch_bounds__(256) void lovasz_keys_kernel(const float *__restrict__ probas, const int64_t *__restrict__ labels,
                                                         int64_t n, int nc, int ncp, int skip, int has_ignore, int64_t ignore,
                                                         uint64_t *__restrict__ keys, uint32_t *__restrict__ vals) {
  extern __shared__ float tile[];
  const int stride = nc | 1;
  const int64_t row0 = (int64_t)blockIdx.x * kRows;
  const int rows = (int)min((int64_t)kRows, n - row0);
  const float *src = probas + row0 * nc;
  for (int e = threadIdx.x; e < rows * nc; e += 256) tile[(e / nc) * stride + (e % nc)] = src[e];
  __syncthreads();
  const int t = threadIdx.x;
  if (t >= rows) return;
  const int64_t i = row0 + t;
  const int64_t l = labels[i];
  const bool valid = label_valid(l, nc, has_ignore, ignore);
  for (int s = 0; s < ncp; ++s) {
    const int c = s + (skip >= 0 && s >= skip ? 1 : 0);
    const bool fg = valid && l == c;
    const float err = valid ? fabsf((fg ? 1.f : 0.f) - tile[t * stride + c]) : 0.f;
    keys[(int64_t)s * n + i] = ((uint64_t)s << 32) | (uint64_t)(0xFFFFFFFFu - __float_as_uint(err));
    vals[(int64_t)s * n + i] = (uint32_t)i | (fg ? 0x80000000u : 0u);
  }
}

__device__ __forceinline__ int block_sum_256(int v, int *lds) {   // every thread gets the sum
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
  if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = v;
  __syncthreads();
  const int r = lds[0] + lds[1] + lds[2] + lds[3];
  __syncthreads();
  return r;
}

__global__ __launch_bounds__(256) void lovasz_blocksum_kernel(const uint32_t *__restrict__ vals, int64_t n, int64_t nblk,
                                                             int32_t *__restrict__ blocksum) {
  __shared__ int lds[4];
  const int s = blockIdx.y;
  const int64_t base = (int64_t)blockIdx.x * kBlockElems;
  const uint32_t *v = vals + (int64_t)s * n;
  int c = 0;
  for (int j = 0; j < kPerThread; ++j) {
    const int64_t e = base + j * 256 + threadIdx.x;
    if (e < n) c += (int)(v[e] >> 31);
  }
  const int tot = block_sum_256(c, lds);
  if (threadIdx.x == 0) blocksum[(int64_t)s * nblk + blockIdx.x] = tot;
}

__device__ __forceinline__ float jaccard(int gts, int64_t pos, int cs) {   // lovasz_losses.py:29-32 at 1-based position pos
  return 1.f - (float)(gts - cs) / (float)((int64_t)gts + pos - cs);
}

__global__ __launch_bounds__(256) void lovasz_final_kernel(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ vals,
                                                          const Header *__restrict__ hdr, const int32_t *__restrict__ blocksum,
                                                          int64_t n, int64_t nblk, int nc, int ncp, int skip,
                                                          float *__restrict__ gradt, double *__restrict__ blockloss) {
  // gradt (may be NULL): the gradient class-slot-major, gradt[s n + point] -- the sorted stream scatters it by point, and a
  // slot's n floats (4.6 MB at 1.16 M points) stay cache-resident while its workgroups run, where the (point, class) layout
  // spread the same writes over the whole 93 MB tensor (0.47 ms); lovasz_grad_rows_kernel turns it into rows afterwards.
  __shared__ int lds[4];
  __shared__ int wave_tot[4];
  __shared__ double dl[4];
  // workgroup -> (class slot, block): every workgroup of a slot on ONE XCD (the dispatcher deals linear workgroup ids round-robin
  // over the 8 XCDs), slot s on XCD s % 8 -- the slot's scattered gradient stores then fill whole lines in one L2 instead of
  // leaving byte-masked pieces of every line in eight of them
  const int64_t lin = blockIdx.x, q = lin >> 3;
  const int s = (int)(lin & 7) + 8 * (int)(q / nblk);
  if (s >= ncp) return;
  const int c = s + (skip >= 0 && s >= skip ? 1 : 0);
  const int64_t b = q % nblk;
  const int t = threadIdx.x;
  const int gts = hdr->cnt[c];
  const int npresent = block_sum_256(t < nc && hdr->cnt[t] > 0 ? 1 : 0, lds);
  const uint64_t *k = keys + (int64_t)s * n;
  const uint32_t *v = vals + (int64_t)s * n;
  const int64_t e0 = b * kBlockElems + (int64_t)t * kPerThread;   // this thread's consecutive elements
  uint32_t val[kPerThread];
  float err[kPerThread];
  int cnt = 0;
#pragma unroll
  for (int j = 0; j < kPerThread; ++j) {
    const bool in = e0 + j < n;
    val[j] = in ? v[e0 + j] : 0u;
    err[j] = in ? __uint_as_float(0xFFFFFFFFu - (uint32_t)k[e0 + j]) : 0.f;
    cnt += (int)(val[j] >> 31);
  }
  if (gts == 0) {   // absent class: no loss term, zero gradient column
    if (gradt != nullptr)
      for (int j = 0; j < kPerThread; ++j)
        if (e0 + j < n) gradt[(int64_t)s * n + (val[j] & 0x7FFFFFFFu)] = 0.f;
    if (t == 0) blockloss[(int64_t)s * nblk + b] = 0.0;
    return;
  }
  // foreground points of this class in the blocks before this one
  int before = 0;
  for (int64_t j = t; j < b; j += 256) before += blocksum[(int64_t)s * nblk + j];
  before = block_sum_256(before, lds);
  // exclusive scan of the per-thread counts over the workgroup
  int incl = cnt;
  for (int o = 1; o < 64; o <<= 1) {
    const int up = __shfl_up(incl, o);
    if ((t & 63) >= o) incl += up;
  }
  if ((t & 63) == 63) wave_tot[t >> 6] = incl;
  __syncthreads();
  int cs = before + incl - cnt;
  for (int w = 0; w < (t >> 6); ++w) cs += wave_tot[w];
  const float inv = 1.f / (float)npresent;
  double acc = 0.0;
  float jprev = e0 > 0 ? jaccard(gts, e0, cs) : 0.f;   // jaccard of the element before this thread's first one
#pragma unroll
  for (int j = 0; j < kPerThread; ++j) {
    if (e0 + j < n) {
      const int fg = (int)(val[j] >> 31);
      cs += fg;
      const float jac = jaccard(gts, e0 + j + 1, cs);
      const float g = jac - jprev;
      jprev = jac;
      acc += (double)err[j] * (double)g;
      if (gradt != nullptr)
        gradt[(int64_t)s * n + (val[j] & 0x7FFFFFFFu)] = err[j] == 0.f ? 0.f : (fg ? -g : g) * inv;
    }
  }
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
  if ((t & 63) == 0) dl[t >> 6] = acc;
  __syncthreads();
  if (t == 0) blockloss[(int64_t)s * nblk + b] = (dl[0] + dl[1]) + (dl[2] + dl[3]);
}

// gradt (slots, n) -> grad (n, nc) rows through the LDS tile of lovasz_keys_kernel; the ignored class's column is zero
__global__ __launch_bounds__(256) void lovasz_grad_rows_kernel(const float *__restrict__ gradt, int64_t n, int nc, int ncp, int skip,
                                                              float *__restrict__ grad) {
  extern __shared__ float tile[];
  const int stride = nc | 1;
  const int64_t row0 = (int64_t)blockIdx.x * kRows;
  const int rows = (int)min((int64_t)kRows, n - row0);
  const int t = threadIdx.x;
  if (t < rows) {
    if (skip >= 0) tile[t * stride + skip] = 0.f;
    for (int s = 0; s < ncp; ++s) {
      const int c = s + (skip >= 0 && s >= skip ? 1 : 0);
      tile[t * stride + c] = gradt[(int64_t)s * n + row0 + t];
    }
  }
  __syncthreads();
  float *dst = grad + row0 * nc;
  for (int e = threadIdx.x; e < rows * nc; e += 256) dst[e] = tile[(e / nc) * stride + (e % nc)];
}

// one wave per class slot (16 waves): lane-strided sums of the slot's block partials, combined in a fixed shuffle order
__global__ __launch_bounds__(1024) void lovasz_reduce_kernel(const Header *__restrict__ hdr, const double *__restrict__ blockloss,
                                                            int64_t nblk, int nc, int ncp, int skip, float *__restrict__ loss) {
  __shared__ double per_class[64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int s = wave; s < ncp; s += 16) {
    const int c = s + (skip >= 0 && s >= skip ? 1 : 0);
    double a = 0.0;
    if (hdr->cnt[c] > 0)
      for (int64_t j = lane; j < nblk; j += 64) a += blockloss[(int64_t)s * nblk + j];
    for (int o = 32; o > 0; o >>= 1) a += __shfl_down(a, o);
    if (lane == 0) per_class[s] = a;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double total = 0.0;
    int npresent = 0;
    for (int s = 0; s < ncp; ++s) {
      const int c = s + (skip >= 0 && s >= skip ? 1 : 0);
      if (hdr->cnt[c] > 0) { total += per_class[s]; ++npresent; }
    }
    loss[0] = npresent > 0 ? (float)(total / (double)npresent) : 0.f;
  }
}

}  // namespace

extern "C" {

int64_t pcs_lovasz_workspace_bytes(int64_t n, int32_t num_class, int32_t has_ignore, int64_t ignore) {
  Plan p;
  if (plan_for(n, num_class, has_ignore, ignore, p) != PCS_OK) return -1;
  return (int64_t)p.bytes;
}

int pcs_lovasz_softmax_f32(const float *probas, const int64_t *labels, int64_t n, int32_t num_class, int32_t has_ignore,
                           int64_t ignore, float *loss, float *grad, void *ws, int64_t ws_bytes, void *stream) {
  Plan p;
  int rc = plan_for(n, num_class, has_ignore, ignore, p);
  if (rc != PCS_OK) return rc;
  if (loss == nullptr || (n > 0 && (probas == nullptr || labels == nullptr))) {
    set_error("pcs_lovasz_softmax_f32: null pointer");
    return PCS_EINVAL;
  }
  hipStream_t st = as_stream(stream);
  if (p.total == 0) {   // no points (or the only class is the ignored one): the loss is an exact zero
    if (hipMemsetAsync(loss, 0, sizeof(float), st) != hipSuccess) return check_launch("pcs_lovasz_softmax_f32");
    if (grad != nullptr && n > 0 && hipMemsetAsync(grad, 0, sizeof(float) * n * num_class, st) != hipSuccess)
      return check_launch("pcs_lovasz_softmax_f32");
    return PCS_OK;
  }
  if (ws == nullptr || ws_bytes < (int64_t)p.bytes) {
    set_error("pcs_lovasz_softmax_f32: workspace of %lld bytes, %lld needed (pcs_lovasz_workspace_bytes)", (long long)ws_bytes,
              (long long)p.bytes);
    return PCS_EINVAL;
  }
  char *base = reinterpret_cast<char *>(ws);
  Header *hdr = reinterpret_cast<Header *>(base);
  int32_t *blocksum = reinterpret_cast<int32_t *>(base + p.off_blocksum);
  double *blockloss = reinterpret_cast<double *>(base + p.off_blockloss);
  rocprim::double_buffer<uint64_t> keys(reinterpret_cast<uint64_t *>(base + p.off_keys[0]),
                                        reinterpret_cast<uint64_t *>(base + p.off_keys[1]));
  rocprim::double_buffer<uint32_t> vals(reinterpret_cast<uint32_t *>(base + p.off_vals[0]),
                                        reinterpret_cast<uint32_t *>(base + p.off_vals[1]));
  if (hipMemsetAsync(hdr, 0, sizeof(Header), st) != hipSuccess) return check_launch("pcs_lovasz_softmax_f32: memset");
  lovasz_count_kernel<<<stream_grid(n, 256), 256, 0, st>>>(labels, n, p.nc, has_ignore, ignore, hdr);
  const int stride = p.nc | 1;
  lovasz_keys_kernel<<<(unsigned)ceil_div(n, kRows), 256, sizeof(float) * kRows * stride, st>>>(
      probas, labels, n, p.nc, p.ncp, p.skip, has_ignore, ignore, keys.current(), vals.current());
  if ((rc = check_launch("lovasz_keys_kernel")) != PCS_OK) return rc;
  size_t temp_bytes = p.temp_bytes;
  hipError_t e = rocprim::radix_sort_pairs(base + p.off_temp, temp_bytes, keys, vals, (size_t)p.total, 0u, (unsigned)p.key_bits, st);
  if (e != hipSuccess) {
    set_error("pcs_lovasz_softmax_f32: rocprim::radix_sort_pairs: %s", hipGetErrorString(e));
    return PCS_ELAUNCH;
  }
  const dim3 grid((unsigned)p.nblk, (unsigned)p.ncp);
  lovasz_blocksum_kernel<<<grid, 256, 0, st>>>(vals.current(), n, p.nblk, blocksum);
  float *gradt = grad != nullptr ? reinterpret_cast<float *>(base + p.off_gradt) : nullptr;
  const int64_t final_grid = 8 * p.nblk * ceil_div(p.ncp, 8);
  if (final_grid > 0x7FFFFFFF) { set_error("pcs_lovasz_softmax_f32: too many work blocks"); return PCS_EINVAL; }
  lovasz_final_kernel<<<(unsigned)final_grid, 256, 0, st>>>(keys.current(), vals.current(), hdr, blocksum, n, p.nblk, p.nc, p.ncp,
                                                            p.skip, gradt, blockloss);
  if (grad != nullptr)
    lovasz_grad_rows_kernel<<<(unsigned)ceil_div(n, kRows), 256, sizeof(float) * kRows * stride, st>>>(gradt, n, p.nc, p.ncp, p.skip, grad);
  lovasz_reduce_kernel<<<1, 1024, 0, st>>>(hdr, blockloss, p.nblk, p.nc, p.ncp, p.skip, loss);
  return check_launch("pcs_lovasz_softmax_f32");
}

}  // extern "C"
