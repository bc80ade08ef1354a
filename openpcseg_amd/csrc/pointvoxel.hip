// K7-K10 point<->voxel kernels + calc_ti_weights -- gfx950, fp32.
// Reference semantics: TS:torchsparse/backend/voxelize/voxelize_cuda.cu:12-80,
// TS:torchsparse/backend/devoxelize/devoxelize_cuda.cu:11-98,
// TS:torchsparse/nn/functional/devoxelize.py:10-48.
// All HBM-bound. The reference launches <<<N, c>>> (c-thread blocks: 4 threads for c = 4) and
// accumulates the 8 trilinear corners through global memory; here a 256-thread workgroup
// covers 256/TX rows with TX lanes x 16 bytes per row, the corner sum lives in registers and
// every output row is written once.
#include "pcs_common.h"

using namespace pcs;

namespace {

// Row-tiled 2-D launch: TX lanes walk the (vectorised) channels of one row, TY rows per block.
struct RowLaunch {
  dim3 block, grid;
  int cv;  // vectors per row
};

template <int V>
RowLaunch row_launch(int64_t n, int c) {
  RowLaunch r;
  r.cv = c / V;
  int tx = 1;
  while (tx < r.cv && tx < 64) tx <<= 1;
  int ty = 256 / tx;
  r.block = dim3(tx, ty);
  int64_t g = ceil_div(n, ty);
  if (g > 256 * 16) g = 256 * 16;
  if (g < 1) g = 1;
  r.grid = dim3((unsigned)g);
  return r;
}

template <int V> struct Vec;
template <> struct Vec<4> { using T = float4; };
template <> struct Vec<1> { using T = float; };

__device__ __forceinline__ float4 vscale(float4 a, float s) { return make_float4(a.x * s, a.y * s, a.z * s, a.w * s); }
__device__ __forceinline__ float vscale(float a, float s) { return a * s; }
__device__ __forceinline__ float4 vdiv(float4 a, float s) { return make_float4(a.x / s, a.y / s, a.z / s, a.w / s); }
__device__ __forceinline__ float vdiv(float a, float s) { return a / s; }
__device__ __forceinline__ float4 vfma(float w, float4 f, float4 a) {
  return make_float4(fmaf(w, f.x, a.x), fmaf(w, f.y, a.y), fmaf(w, f.z, a.z), fmaf(w, f.w, a.w));
}
__device__ __forceinline__ float vfma(float w, float f, float a) { return fmaf(w, f, a); }
__device__ __forceinline__ void vatomic_add(float *p, float4 v) {
  atomicAdd(p + 0, v.x); atomicAdd(p + 1, v.y); atomicAdd(p + 2, v.z); atomicAdd(p + 3, v.w);
}
__device__ __forceinline__ void vatomic_add(float *p, float v) { atomicAdd(p, v); }
__device__ __forceinline__ void vzero(float4 &v) { v = make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ void vzero(float &v) { v = 0.f; }

// ---- K7: scatter-mean -----------------------------------------------------------------------
template <int V>
__global__ void __launch_bounds__(256) voxelize_fwd_kernel(const float *__restrict__ feats,
                                                           const int32_t *__restrict__ idx,
                                                           const int32_t *__restrict__ counts,
                                                           int64_t n, int64_t m, int c, int cv,
                                                           float *out) {
  using VT = typename Vec<V>::T;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.y + threadIdx.y; i < n;
       i += (int64_t)gridDim.x * blockDim.y) {
    const int32_t pos = idx[i];
    if (pos < 0 || pos >= m) continue;
    const int32_t cnt = counts[pos];
    if (cnt == 0) continue;
    const float fc = (float)cnt;
    const VT *src = reinterpret_cast<const VT *>(feats + i * c);
    float *dst = out + (int64_t)pos * c;
    for (int j = threadIdx.x; j < cv; j += blockDim.x) vatomic_add(dst + j * V, vdiv(src[j], fc));
  }
}

// ---- K7, contention-free form: per-voxel segmented mean over a CSR of the points (order = point rows sorted by
// voxel, rowptr (m+1)). Every output row is written exactly once: no memset, no atomics, deterministic; the
// atomic form above runs at 0.55 TB/s on 1.4 M points (random voxel rows), this one at gather speed.
template <int V>
__global__ void __launch_bounds__(256) voxelize_fwd_csr_kernel(const float *__restrict__ feats,
                                                               const int64_t *__restrict__ order,
                                                               const int64_t *__restrict__ rowptr,
                                                               const int32_t *__restrict__ counts,
                                                               int64_t m, int c, int cv, float *__restrict__ out) {
  using VT = typename Vec<V>::T;
  for (int64_t v = (int64_t)blockIdx.x * blockDim.y + threadIdx.y; v < m;
       v += (int64_t)gridDim.x * blockDim.y) {
    const int64_t e0 = rowptr[v], e1 = rowptr[v + 1];
    const int32_t cnt = counts[v];
    const float fc = (float)(cnt > 0 ? cnt : 1);
    VT *dst = reinterpret_cast<VT *>(out + v * c);
    for (int j = threadIdx.x; j < cv; j += blockDim.x) {
      VT acc; vzero(acc);
      if (cnt != 0) {
        int64_t e = e0;
        for (; e + 1 < e1; e += 2) {  // two independent row loads in flight
          const int64_t p0 = order[e], p1 = order[e + 1];
          const VT f0 = vdiv(reinterpret_cast<const VT *>(feats + p0 * c)[j], fc);  // divide, then add
          const VT f1 = vdiv(reinterpret_cast<const VT *>(feats + p1 * c)[j], fc);  // (voxelize_cuda.cu:27-29)
          acc = vfma(1.f, f0, acc);
          acc = vfma(1.f, f1, acc);
        }
        if (e < e1) acc = vfma(1.f, vdiv(reinterpret_cast<const VT *>(feats + order[e] * c)[j], fc), acc);
      }
      dst[j] = acc;
    }
  }
}

// ---- K8: gather back / count ------------------------------------------------------------------
template <int V>
__global__ void __launch_bounds__(256) voxelize_bwd_kernel(const float *__restrict__ gout,
                                                           const int32_t *__restrict__ idx,
                                                           const int32_t *__restrict__ counts,
                                                           int64_t n, int c, int cv,
                                                           float *__restrict__ gin) {
  using VT = typename Vec<V>::T;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.y + threadIdx.y; i < n;
       i += (int64_t)gridDim.x * blockDim.y) {
    const int32_t pos = idx[i];
    VT *dst = reinterpret_cast<VT *>(gin + i * c);
    int32_t cnt = pos >= 0 ? counts[pos] : 0;
    if (cnt == 0) {
      VT z; vzero(z);
      for (int j = threadIdx.x; j < cv; j += blockDim.x) dst[j] = z;
      continue;
    }
    const float fc = (float)cnt;
    const VT *src = reinterpret_cast<const VT *>(gout + (int64_t)pos * c);
    for (int j = threadIdx.x; j < cv; j += blockDim.x) dst[j] = vdiv(src[j], fc);
  }
}

// ---- K9: trilinear gather ---------------------------------------------------------------------
template <int V>
__global__ void __launch_bounds__(256) devoxelize_fwd_kernel(const float *__restrict__ feat,
                                                             const int32_t *__restrict__ idx8,
                                                             const float *__restrict__ w8,
                                                             int64_t n, int c, int cv,
                                                             float *__restrict__ out) {
  using VT = typename Vec<V>::T;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.y + threadIdx.y; i < n;
       i += (int64_t)gridDim.x * blockDim.y) {
    int32_t id[8];
    float w[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { id[k] = idx8[i * 8 + k]; w[k] = w8[i * 8 + k]; }
    VT *dst = reinterpret_cast<VT *>(out + i * c);
    for (int j = threadIdx.x; j < cv; j += blockDim.x) {
      VT acc; vzero(acc);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (id[k] >= 0) acc = vfma(w[k], reinterpret_cast<const VT *>(feat + (int64_t)id[k] * c)[j], acc);
      }
      dst[j] = acc;
    }
  }
}

// ---- K10: trilinear scatter -------------------------------------------------------------------
template <int V>
__global__ void __launch_bounds__(256) devoxelize_bwd_kernel(const float *__restrict__ gout,
                                                             const int32_t *__restrict__ idx8,
                                                             const float *__restrict__ w8,
                                                             int64_t n, int c, int cv,
                                                             float *gfeat) {
  using VT = typename Vec<V>::T;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.y + threadIdx.y; i < n;
       i += (int64_t)gridDim.x * blockDim.y) {
    int32_t id[8];
    float w[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { id[k] = idx8[i * 8 + k]; w[k] = w8[i * 8 + k]; }
    const VT *src = reinterpret_cast<const VT *>(gout + i * c);
    for (int j = threadIdx.x; j < cv; j += blockDim.x) {
      const VT g = src[j];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (id[k] >= 0) vatomic_add(gfeat + (int64_t)id[k] * c + j * V, vscale(g, w[k]));
      }
    }
  }
}

// ---- calc_ti_weights ----------------------------------------------------------------------------
__global__ void __launch_bounds__(256) ti_weights_kernel(const float *__restrict__ coords, int ld,
                                                         const int64_t *__restrict__ idxq,
                                                         int64_t n, float scale, float inv_s3,
                                                         int scaled, float *__restrict__ w) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const float x = coords[i * ld + 0], y = coords[i * ld + 1], z = coords[i * ld + 2];
    float xf, yf, zf;
    if (scaled) {
      xf = floorf(x / scale) * scale; yf = floorf(y / scale) * scale; zf = floorf(z / scale) * scale;
    } else {
      xf = floorf(x); yf = floorf(y); zf = floorf(z);
    }
    const float xc = xf + scale, yc = yf + scale, zc = zf + scale;
    const float ax[2] = {xc - x, x - xf}, ay[2] = {yc - y, y - yf}, az[2] = {zc - z, z - zf};
    float wk[8];
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {  // corner order = get_kernel_offsets(2): z fastest
      float v = (ax[(k >> 2) & 1] * ay[(k >> 1) & 1]) * az[k & 1];
      if (scaled) v *= inv_s3;
      if (idxq[(int64_t)k * n + i] == -1) v = 0.f;
      wk[k] = v;
      sum += v;
    }
    const float den = sum + 1e-8f;
#pragma unroll
    for (int k = 0; k < 8; ++k) w[(int64_t)k * n + i] = wk[k] / den;
  }
}

// ---- voxel_to_point map in one pass -------------------------------------------------------
// What R:pcseg/model/segmentor/voxel/minkunet/utils.py:69-105 builds with floor / cat / K2 (8 hashes per point) /
// hashquery (table rebuilt) / calc_ti_weights (~25 kernels) / two transposes: for every point the rows of the 8
// corner voxels of its stride-s cell (-1 = absent) and the trilinear weights, already in the (N,8) layout K9 reads.
// Corner order = get_kernel_offsets(2, s): z fastest. Weights in the reference's fp32 op order (ti_weights_kernel).
__global__ void __launch_bounds__(256) corner_map_kernel(const float *__restrict__ coords, int ld, int64_t n, int stride,
                                                         TableView table, int32_t *__restrict__ idx8,
                                                         float *__restrict__ w8) {
  const float scale = (float)stride;
  const float inv_s3 = 1.0f / (scale * scale * scale);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float x = coords[i * ld + 0], y = coords[i * ld + 1], z = coords[i * ld + 2];
    const int b = (int)coords[i * ld + ld - 1];
    const int bx = (int)floorf(x / scale) * stride, by = (int)floorf(y / scale) * stride, bz = (int)floorf(z / scale) * stride;
    float xf, yf, zf;
    if (stride != 1) {
      xf = floorf(x / scale) * scale; yf = floorf(y / scale) * scale; zf = floorf(z / scale) * scale;
    } else {
      xf = floorf(x); yf = floorf(y); zf = floorf(z);
    }
    const float xc = xf + scale, yc = yf + scale, zc = zf + scale;
    const float ax[2] = {xc - x, x - xf}, ay[2] = {yc - y, y - yf}, az[2] = {zc - z, z - zf};
    float wk[8];
    int id[8];
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int ix = (k >> 2) & 1, iy = (k >> 1) & 1, iz = k & 1;
      id[k] = table_lookup(table, fnv60(bx + ix * stride, by + iy * stride, bz + iz * stride, b));
      float v = (ax[ix] * ay[iy]) * az[iz];
      if (stride != 1) v *= inv_s3;
      if (id[k] < 0) v = 0.f;
      wk[k] = v;
      sum += v;
    }
    const float den = sum + 1e-8f;
    int4 *io = reinterpret_cast<int4 *>(idx8 + i * 8);
    float4 *wo = reinterpret_cast<float4 *>(w8 + i * 8);
    io[0] = make_int4(id[0], id[1], id[2], id[3]);
    io[1] = make_int4(id[4], id[5], id[6], id[7]);
    wo[0] = make_float4(wk[0] / den, wk[1] / den, wk[2] / den, wk[3] / den);
    wo[1] = make_float4(wk[4] / den, wk[5] / den, wk[6] / den, wk[7] / den);
  }
}

// ---- K10, contention-free form: per-voxel segmented reduction over a CSR of the (point, corner)
// entries. entries sorted by voxel: order[e] = flat index i*8+k into idx8/w8; rowptr (m+1).
// One row of TX lanes per voxel; every gfeat row is written exactly once (no memset, no atomics,
// deterministic). The reference's atomicAdd form (devoxelize_cuda.cu:37-57) serialises badly when
// thousands of points share a coarse voxel (stride 16: ~250 entries per voxel).
template <int V>
__global__ void __launch_bounds__(256) devoxelize_bwd_csr_kernel(const float *__restrict__ gout,
                                                                 const int64_t *__restrict__ order,
                                                                 const int64_t *__restrict__ rowptr,
                                                                 const float *__restrict__ w8,
                                                                 int64_t m, int c, int cv,
                                                                 float *__restrict__ gfeat) {
  using VT = typename Vec<V>::T;
  for (int64_t v = (int64_t)blockIdx.x * blockDim.y + threadIdx.y; v < m;
       v += (int64_t)gridDim.x * blockDim.y) {
    const int64_t e0 = rowptr[v], e1 = rowptr[v + 1];
    VT *dst = reinterpret_cast<VT *>(gfeat + v * c);
    for (int j = threadIdx.x; j < cv; j += blockDim.x) {
      VT acc; vzero(acc);
      int64_t e = e0;
      for (; e + 1 < e1; e += 2) {  // two independent row loads in flight
        const int64_t p0 = order[e], p1 = order[e + 1];
        const float w0 = w8[p0], w1 = w8[p1];
        const VT g0 = reinterpret_cast<const VT *>(gout + (p0 >> 3) * c)[j];
        const VT g1 = reinterpret_cast<const VT *>(gout + (p1 >> 3) * c)[j];
        acc = vfma(w0, g0, acc);
        acc = vfma(w1, g1, acc);
      }
      if (e < e1) {
        const int64_t p0 = order[e];
        acc = vfma(w8[p0], reinterpret_cast<const VT *>(gout + (p0 >> 3) * c)[j], acc);
      }
      dst[j] = acc;
    }
  }
}

// The same reduction for NARROW rows (c <= 32, 16-byte granular: the class scores the workload devoxelises since round 4):
// one WAVE per voxel, 8 x-lanes over the row's float4 pieces times 8 entry lanes striding over the voxel's segment, combined
// by shuffles in a fixed order (deterministic). A coarse voxel's ~250 entries are 32 trips of 8 independent row loads instead
// of 125 trips of two in one thread row, and a stride-1 voxel's 8 entries are one trip.
__global__ void __launch_bounds__(256) devoxelize_bwd_csr_narrow_kernel(const float *__restrict__ gout,
                                                                        const int64_t *__restrict__ order,
                                                                        const int64_t *__restrict__ rowptr,
                                                                        const float *__restrict__ w8, int64_t m, int c, int cv,
                                                                        float *__restrict__ gfeat) {
  const int x = threadIdx.x & 7, el = (threadIdx.x >> 3) & 7, wv = threadIdx.x >> 6;
  for (int64_t v = (int64_t)blockIdx.x * 4 + wv; v < m; v += (int64_t)gridDim.x * 4) {   // (uniform over the wave)
    const int64_t e0 = rowptr[v], e1 = rowptr[v + 1];
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t e = e0 + el; e < e1; e += 16) {   // two independent row loads in flight per lane
      const int64_t p0 = order[e];
      const bool two = e + 8 < e1;
      const int64_t p1 = two ? order[e + 8] : p0;
      const float w0 = w8[p0], w1 = w8[p1];
      if (x < cv) {
        const float4 g0 = reinterpret_cast<const float4 *>(gout + (p0 >> 3) * c)[x];
        const float4 g1 = reinterpret_cast<const float4 *>(gout + (p1 >> 3) * c)[x];
        acc = vfma(w0, g0, acc);
        if (two) acc = vfma(w1, g1, acc);   // (not a zero weight: 0 x Inf of a diverged gradient must not turn into NaN here)
      }
    }
#pragma unroll
    for (int o = 32; o >= 8; o >>= 1) {
      acc.x += __shfl_down(acc.x, o, 64); acc.y += __shfl_down(acc.y, o, 64);
      acc.z += __shfl_down(acc.z, o, 64); acc.w += __shfl_down(acc.w, o, 64);
    }
    if (el == 0 && x < cv) reinterpret_cast<float4 *>(gfeat + v * c)[x] = acc;
  }
}

}  // namespace

static bool aligned16(const void *p) { return ((uintptr_t)p & 15) == 0; }

extern "C" int pcs_voxelize_fwd_f32(const float *feats, const int32_t *idx,
                                    const int32_t *counts, int64_t n, int64_t m, int32_t c,
                                    float *out, void *stream) {
  if (n < 0 || m < 0 || c <= 0) { set_error("pcs_voxelize_fwd: bad sizes"); return PCS_EINVAL; }
  hipStream_t st = as_stream(stream);
  if (m > 0) {
    if (!out) { set_error("pcs_voxelize_fwd: null out"); return PCS_EINVAL; }
    if (hipMemsetAsync(out, 0, (size_t)m * c * 4, st) != hipSuccess) { set_error("pcs_voxelize_fwd: memset failed"); return PCS_ELAUNCH; }
  }
  if (n == 0 || m == 0) return PCS_OK;
  if (!feats || !idx || !counts) { set_error("pcs_voxelize_fwd: null input"); return PCS_EINVAL; }
  if ((c & 3) == 0 && aligned16(feats)) {
    RowLaunch rl = row_launch<4>(n, c);
    hipLaunchKernelGGL(voxelize_fwd_kernel<4>, rl.grid, rl.block, 0, st, feats, idx, counts, n, m, c, rl.cv, out);
  } else {
    RowLaunch rl = row_launch<1>(n, c);
    hipLaunchKernelGGL(voxelize_fwd_kernel<1>, rl.grid, rl.block, 0, st, feats, idx, counts, n, m, c, rl.cv, out);
  }
  return check_launch("pcs_voxelize_fwd");
}

extern "C" int pcs_voxelize_fwd_csr_f32(const float *feats, const int64_t *order, const int64_t *rowptr,
                                        const int32_t *counts, int64_t m, int32_t c, float *out, void *stream) {
  if (m < 0 || c <= 0) { set_error("pcs_voxelize_fwd_csr: bad sizes"); return PCS_EINVAL; }
  if (m == 0) return PCS_OK;
  if (!order || !rowptr || !counts || !out) { set_error("pcs_voxelize_fwd_csr: null pointer"); return PCS_EINVAL; }
  hipStream_t st = as_stream(stream);
  if ((c & 3) == 0 && aligned16(feats) && aligned16(out)) {
    RowLaunch rl = row_launch<4>(m, c);
    hipLaunchKernelGGL(voxelize_fwd_csr_kernel<4>, rl.grid, rl.block, 0, st, feats, order, rowptr, counts, m, c, rl.cv, out);
  } else {
    RowLaunch rl = row_launch<1>(m, c);
    hipLaunchKernelGGL(voxelize_fwd_csr_kernel<1>, rl.grid, rl.block, 0, st, feats, order, rowptr, counts, m, c, rl.cv, out);
  }
  return check_launch("pcs_voxelize_fwd_csr");
}

extern "C" int pcs_voxelize_bwd_f32(const float *gout, const int32_t *idx, const int32_t *counts,
                                    int64_t n, int32_t c, float *gin, void *stream) {
  if (n < 0 || c <= 0) { set_error("pcs_voxelize_bwd: bad sizes"); return PCS_EINVAL; }
  if (n == 0) return PCS_OK;
  if (!gout || !idx || !counts || !gin) { set_error("pcs_voxelize_bwd: null pointer"); return PCS_EINVAL; }
  hipStream_t st = as_stream(stream);
  if ((c & 3) == 0 && aligned16(gout) && aligned16(gin)) {
    RowLaunch rl = row_launch<4>(n, c);
    hipLaunchKernelGGL(voxelize_bwd_kernel<4>, rl.grid, rl.block, 0, st, gout, idx, counts, n, c, rl.cv, gin);
  } else {
    RowLaunch rl = row_launch<1>(n, c);
    hipLaunchKernelGGL(voxelize_bwd_kernel<1>, rl.grid, rl.block, 0, st, gout, idx, counts, n, c, rl.cv, gin);
  }
  return check_launch("pcs_voxelize_bwd");
}

extern "C" int pcs_devoxelize_fwd_f32(const float *feat, const int32_t *idx8, const float *w8,
                                      int64_t n, int32_t c, float *out, void *stream) {
  if (n < 0 || c <= 0) { set_error("pcs_devoxelize_fwd: bad sizes"); return PCS_EINVAL; }
  if (n == 0) return PCS_OK;
  if (!idx8 || !w8 || !out) { set_error("pcs_devoxelize_fwd: null pointer"); return PCS_EINVAL; }
  hipStream_t st = as_stream(stream);
  if ((c & 3) == 0 && aligned16(feat) && aligned16(out)) {
    RowLaunch rl = row_launch<4>(n, c);
    hipLaunchKernelGGL(devoxelize_fwd_kernel<4>, rl.grid, rl.block, 0, st, feat, idx8, w8, n, c, rl.cv, out);
  } else {
    RowLaunch rl = row_launch<1>(n, c);
    hipLaunchKernelGGL(devoxelize_fwd_kernel<1>, rl.grid, rl.block, 0, st, feat, idx8, w8, n, c, rl.cv, out);
  }
  return check_launch("pcs_devoxelize_fwd");
}

extern "C" int pcs_devoxelize_bwd_f32(const float *gout, const int32_t *idx8, const float *w8,
                                      int64_t n, int64_t m, int32_t c, float *gfeat,
                                      void *stream) {
  if (n < 0 || m < 0 || c <= 0) { set_error("pcs_devoxelize_bwd: bad sizes"); return PCS_EINVAL; }
  hipStream_t st = as_stream(stream);
  if (m > 0) {
    if (!gfeat) { set_error("pcs_devoxelize_bwd: null gfeat"); return PCS_EINVAL; }
    if (hipMemsetAsync(gfeat, 0, (size_t)m * c * 4, st) != hipSuccess) { set_error("pcs_devoxelize_bwd: memset failed"); return PCS_ELAUNCH; }
  }
  if (n == 0 || m == 0) return PCS_OK;
  if (!gout || !idx8 || !w8) { set_error("pcs_devoxelize_bwd: null pointer"); return PCS_EINVAL; }
  if ((c & 3) == 0 && aligned16(gout)) {
    RowLaunch rl = row_launch<4>(n, c);
    hipLaunchKernelGGL(devoxelize_bwd_kernel<4>, rl.grid, rl.block, 0, st, gout, idx8, w8, n, c, rl.cv, gfeat);
  } else {
    RowLaunch rl = row_launch<1>(n, c);
    hipLaunchKernelGGL(devoxelize_bwd_kernel<1>, rl.grid, rl.block, 0, st, gout, idx8, w8, n, c, rl.cv, gfeat);
  }
  return check_launch("pcs_devoxelize_bwd");
}

extern "C" int pcs_ti_weights_f32(const float *coords, int32_t coord_ld, const int64_t *idx_query,
                                  int64_t n, float scale, float *w, void *stream) {
  if (n < 0 || coord_ld < 3) { set_error("pcs_ti_weights: bad sizes"); return PCS_EINVAL; }
  if (n == 0) return PCS_OK;
  if (!coords || !idx_query || !w) { set_error("pcs_ti_weights: null pointer"); return PCS_EINVAL; }
  const int scaled = (scale != 1.0f);
  const float inv_s3 = 1.0f / (scale * scale * scale);
  hipLaunchKernelGGL(ti_weights_kernel, dim3(stream_grid(n, 256)), dim3(256), 0, as_stream(stream),
                     coords, coord_ld, idx_query, n, scale, inv_s3, scaled, w);
  return check_launch("pcs_ti_weights");
}

extern "C" int pcs_corner_map_f32(const float *coords, int32_t coord_ld, int64_t n, int32_t stride, const void *table,
                                  int64_t capacity, int32_t *idx8, float *w8, void *stream) {
  if (n < 0 || coord_ld < 4 || stride <= 0 || !table || capacity <= 0 || (capacity & (capacity - 1))) {
    set_error("pcs_corner_map_f32: bad args");
    return PCS_EINVAL;
  }
  if (n == 0) return PCS_OK;
  if (!coords || !idx8 || !w8 || !aligned16(idx8) || !aligned16(w8)) { set_error("pcs_corner_map_f32: bad pointers"); return PCS_EINVAL; }
  hipLaunchKernelGGL(corner_map_kernel, dim3(stream_grid(n, 256)), dim3(256), 0, as_stream(stream), coords, coord_ld, n,
                     stride, make_view(table, capacity), idx8, w8);
  return check_launch("pcs_corner_map_f32");
}

extern "C" int pcs_devoxelize_bwd_csr_f32(const float *gout, const int64_t *order,
                                          const int64_t *rowptr, const float *w8, int64_t m,
                                          int32_t c, float *gfeat, void *stream) {
  if (m < 0 || c <= 0) { set_error("pcs_devoxelize_bwd_csr: bad sizes"); return PCS_EINVAL; }
  if (m == 0) return PCS_OK;
  if (!gout || !order || !rowptr || !w8 || !gfeat) { set_error("pcs_devoxelize_bwd_csr: null pointer"); return PCS_EINVAL; }
  hipStream_t st = as_stream(stream);
  // wave-per-voxel form: long segments (coarse levels). On a stride-1 level (~8 entries per voxel, m ~ 1 M) most of its entry
  // lanes idle and the row-per-thread-row form is 2x faster [r4: 257 vs 117 us]; m is the proxy for the segment length here
  // (the entry count is not an argument of this call)
  if ((c & 3) == 0 && c <= 32 && m <= 400000 && aligned16(gout) && aligned16(gfeat)) {
    int64_t g = ceil_div(m, 4);
    if (g > 256 * 64) g = 256 * 64;
    hipLaunchKernelGGL(devoxelize_bwd_csr_narrow_kernel, dim3((unsigned)g), dim3(256), 0, st, gout, order, rowptr, w8, m, c, c / 4, gfeat);
  } else if ((c & 3) == 0 && aligned16(gout) && aligned16(gfeat)) {
    RowLaunch rl = row_launch<4>(m, c);
    hipLaunchKernelGGL(devoxelize_bwd_csr_kernel<4>, rl.grid, rl.block, 0, st, gout, order, rowptr, w8, m, c, rl.cv, gfeat);
  } else {
    RowLaunch rl = row_launch<1>(m, c);
    hipLaunchKernelGGL(devoxelize_bwd_csr_kernel<1>, rl.grid, rl.block, 0, st, gout, order, rowptr, w8, m, c, rl.cv, gfeat);
  }
  return check_launch("pcs_devoxelize_bwd_csr");
}
