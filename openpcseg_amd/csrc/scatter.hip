// Cylindrical scatter (torch_scatter.scatter_max / scatter_mean as used by the reference's cylinder front-end,
// R:tools/utils/common/seg_utils.py:172-188, R:pcseg/model/segmentor/voxel/cylinder3d/cylinder_ts.py:24-43) and the
// range-image scatter of RPVNet (K13-K15, RL:range_utils/src/map_count_gpu.cu:5-14, denselize_gpu.cu:5-34) -- gfx950.
// All HBM-bound. scatter_max runs as a per-voxel segmented reduction over a sorted CSR (deterministic, no float
// atomic-max tricks, argmax = first point in CSR order); the range scatter runs tiled over a CSR of the pixels
// (full-line NCHW accesses, no atomics); the reference's atomic dataflow is kept for odd channel counts and A/B.
#include "pcs_common.h"

using namespace pcs;

namespace {

struct RowGrid { dim3 block, grid; };
RowGrid row_grid(int64_t rows, int c) {
  int tx = 1;
  while (tx < c && tx < 64) tx <<= 1;
  RowGrid r;
  r.block = dim3(tx, 256 / tx);
  int64_t g = ceil_div(rows > 0 ? rows : 1, 256 / tx);
  if (g > 256 * 16) g = 256 * 16;
  r.grid = dim3((unsigned)g);
  return r;
}

// out[v, j] = max over the points of voxel v of src[i, j]; arg[v, j] = that point (first in CSR order on ties);
// empty voxels: out = 0, arg = -1. V = 4: one lane owns 4 channels (16-byte loads, 16-byte stores of out and arg).
// arg is int32 (point rows < 2^31): an int64 argmax would be half of the kernel's HBM traffic.
template <int V> struct SV;
template <> struct SV<4> { using F = float4; using I = int4; };
template <> struct SV<1> { using F = float; using I = int; };
__device__ __forceinline__ float sget(const float4 &v, int q) { return q == 0 ? v.x : (q == 1 ? v.y : (q == 2 ? v.z : v.w)); }
__device__ __forceinline__ float sget(const float &v, int) { return v; }
__device__ __forceinline__ float4 spack(const float *a, float4 *) { return make_float4(a[0], a[1], a[2], a[3]); }
__device__ __forceinline__ float spack(const float *a, float *) { return a[0]; }
__device__ __forceinline__ int4 spack(const int *a, int4 *) { return make_int4(a[0], a[1], a[2], a[3]); }
__device__ __forceinline__ int spack(const int *a, int *) { return a[0]; }

template <int V>
__global__ void __launch_bounds__(256) scatter_max_csr_kernel(const float *__restrict__ src,
                                                              const int64_t *__restrict__ order,
                                                              const int64_t *__restrict__ rowptr, int64_t m,
                                                              int c, int cv, float *__restrict__ out,
                                                              int32_t *__restrict__ arg) {
  using F = typename SV<V>::F;
  using I = typename SV<V>::I;
  for (int64_t v = (int64_t)blockIdx.x * blockDim.y + threadIdx.y; v < m; v += (int64_t)gridDim.x * blockDim.y) {
    const int64_t e0 = rowptr[v], e1 = rowptr[v + 1];
    for (int j = threadIdx.x; j < cv; j += blockDim.x) {
      float best[V];
      int bi[V];
#pragma unroll
      for (int q = 0; q < V; ++q) { best[q] = 0.f; bi[q] = -1; }
      for (int64_t e = e0; e < e1; ++e) {
        const int64_t i = order[e];
        const F x = reinterpret_cast<const F *>(src + i * c)[j];
#pragma unroll
        for (int q = 0; q < V; ++q) {
          const float xq = sget(x, q);
          if (bi[q] < 0 || xq > best[q]) { best[q] = xq; bi[q] = (int)i; }
        }
      }
      reinterpret_cast<F *>(out + v * c)[j] = spack(best, (F *)nullptr);
      reinterpret_cast<I *>(arg + v * c)[j] = spack(bi, (I *)nullptr);
    }
  }
}

// grad_src is zero-filled by the caller of the C entry point; every (v, j) routes to exactly one element
__global__ void __launch_bounds__(256) scatter_max_bwd_kernel(const float *__restrict__ gout,
                                                              const int32_t *__restrict__ arg, int64_t m, int c,
                                                              float *__restrict__ gsrc) {
  for (int64_t v = (int64_t)blockIdx.x * blockDim.y + threadIdx.y; v < m; v += (int64_t)gridDim.x * blockDim.y) {
    for (int j = threadIdx.x; j < c; j += blockDim.x) {
      const int64_t i = arg[v * c + j];
      if (i >= 0) gsrc[i * c + j] = gout[v * c + j];
    }
  }
}

__global__ void __launch_bounds__(256) map_count_kernel(const int32_t *__restrict__ pxpy, int64_t n, int B, int H,
                                                        int W, int32_t *out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int b = pxpy[3 * i], px = pxpy[3 * i + 1], py = pxpy[3 * i + 2];
    if (b >= 0 && b < B && px >= 0 && px < W && py >= 0 && py < H) atomicAdd(&out[((int64_t)b * H + py) * W + px], 1);
  }
}

__global__ void __launch_bounds__(256) denselize_fwd_kernel(const float *__restrict__ feat,
                                                            const int32_t *__restrict__ cnt,
                                                            const int32_t *__restrict__ pxpy, int64_t n, int B, int C,
                                                            int H, int W, float *out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.y + threadIdx.y; i < n; i += (int64_t)gridDim.x * blockDim.y) {
    const int b = pxpy[3 * i], px = pxpy[3 * i + 1], py = pxpy[3 * i + 2];
    if (b < 0 || b >= B || px < 0 || px >= W || py < 0 || py >= H) continue;
    const int64_t pos = ((int64_t)b * H + py) * W + px;
    const int cm = cnt[pos];
    if (cm == 0) continue;
    const float inv = (float)cm;
    for (int j = threadIdx.x; j < C; j += blockDim.x)
      atomicAdd(&out[(((int64_t)b * C + j) * H + py) * W + px], feat[i * C + j] / inv);
  }
}

// Contention-free point <-> range-image mean. A workgroup owns 64 consecutive pixels of one image row and a chunk
// of CH (32 or 64) channels, staged through an LDS tile [pixel][channel]:
//   forward : per pixel, the feature rows of its points (CSR over pixels: order / rowptr) are summed by CH/4 lanes
//             with 16-byte row loads -- 256 / (CH/4) pixels in flight per workgroup, the dependent
//             rowptr -> order -> row chain is the latency to hide -- then the tile is written transposed, lanes
//             over pixels, so that every NCHW store is a full 256-byte line. Each output element is written exactly
//             once (zeros for empty pixels): no memset, no atomics, deterministic.
//   backward: the NCHW gradient tile is loaded in full lines, then every point of every pixel gets its row
//             (tile row / count) with 16-byte stores; rows of out-of-image points stay zero (memset by the entry).
// The reference's dataflow (RL:range_utils/src/denselize_gpu.cu:5-34: one 4-byte access per element at a H*W
// stride, atomic in forward) measured 2.2 ms forward / 0.85 ms backward on 1.4 M points x 32 channels.
constexpr int DN_PX = 64;
template <int CH, bool BWD>
__global__ void __launch_bounds__(256) denselize_csr_kernel(const float *__restrict__ feat,  // fwd: in ; bwd: unused
                                                            float *__restrict__ gfeat,       // bwd: out
                                                            const int64_t *__restrict__ order,
                                                            const int64_t *__restrict__ rowptr,
                                                            const int32_t *__restrict__ cnt, int B, int C, int H, int W,
                                                            float *__restrict__ img) {        // fwd: out ; bwd: in (gout)
  constexpr int VL = CH / 4;          // lanes per pixel
  constexpr int PP = 256 / VL;        // pixels in flight
  __shared__ __attribute__((aligned(16))) float tile[DN_PX][CH + 4];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int xt = (W + DN_PX - 1) / DN_PX;
  const int x0 = (blockIdx.x % xt) * DN_PX;
  const int64_t row = blockIdx.x / xt;  // b * H + y
  const int c0 = blockIdx.y * CH;
  const int b = (int)(row / H), y = (int)(row % H);
  const int vl = threadIdx.x % VL, p0 = threadIdx.x / VL;
  const int ch = c0 + 4 * vl;
  if (BWD) {
    if (x0 + lane < W)
      for (int j = wid; j < CH; j += 4)
        tile[lane][j] = c0 + j < C ? img[(((int64_t)b * C + c0 + j) * H + y) * W + x0 + lane] : 0.f;
    __syncthreads();
  }
  for (int p = p0; p < DN_PX; p += PP) {
    const int x = x0 + p;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (x < W && ch < C) {
      const int64_t pos = row * W + x;
      const int cm = cnt[pos];
      if (cm != 0) {
        const float fc = (float)cm;
        const int64_t e1 = rowptr[pos + 1];
        if (BWD) {
          const float4 g = *reinterpret_cast<const float4 *>(&tile[p][4 * vl]);
          const float4 o = make_float4(g.x / fc, g.y / fc, g.z / fc, g.w / fc);
          for (int64_t e = rowptr[pos]; e < e1; ++e) *reinterpret_cast<float4 *>(gfeat + order[e] * C + ch) = o;
        } else {
          for (int64_t e = rowptr[pos]; e < e1; ++e) {  // divide, then add (denselize_gpu.cu:14-16)
            const float4 f = *reinterpret_cast<const float4 *>(feat + order[e] * C + ch);
            acc.x += f.x / fc; acc.y += f.y / fc; acc.z += f.z / fc; acc.w += f.w / fc;
          }
        }
      }
    }
    if (!BWD) *reinterpret_cast<float4 *>(&tile[p][4 * vl]) = acc;
  }
  if (!BWD) {
    __syncthreads();
    if (x0 + lane < W)
      for (int j = wid; j < CH && c0 + j < C; j += 4)  // wave per channel, lanes over pixels
        img[(((int64_t)b * C + c0 + j) * H + y) * W + x0 + lane] = tile[lane][j];
  }
}

__global__ void __launch_bounds__(256) denselize_bwd_kernel(const float *__restrict__ gout,
                                                            const int32_t *__restrict__ cnt,
                                                            const int32_t *__restrict__ pxpy, int64_t n, int B, int C,
                                                            int H, int W, float *__restrict__ gfeat) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.y + threadIdx.y; i < n; i += (int64_t)gridDim.x * blockDim.y) {
    const int b = pxpy[3 * i], px = pxpy[3 * i + 1], py = pxpy[3 * i + 2];
    const bool ok = b >= 0 && b < B && px >= 0 && px < W && py >= 0 && py < H;
    const int cm = ok ? cnt[((int64_t)b * H + py) * W + px] : 0;
    for (int j = threadIdx.x; j < C; j += blockDim.x)
      gfeat[i * C + j] = cm > 0 ? gout[(((int64_t)b * C + j) * H + py) * W + px] / (float)cm : 0.f;
  }
}

}  // namespace

extern "C" int pcs_scatter_max_fwd_f32(const float *src, const int64_t *order, const int64_t *rowptr, int64_t m,
                                       int32_t c, float *out, int32_t *arg, void *stream) {
  if (m < 0 || c <= 0) { set_error("pcs_scatter_max_fwd: bad sizes"); return PCS_EINVAL; }
  if (m == 0) return PCS_OK;
  if (!src || !order || !rowptr || !out || !arg) { set_error("pcs_scatter_max_fwd: null pointer"); return PCS_EINVAL; }
  if ((c & 3) == 0 && (((uintptr_t)src | (uintptr_t)out | (uintptr_t)arg) & 15) == 0) {
    RowGrid rg = row_grid(m, c / 4);
    hipLaunchKernelGGL(scatter_max_csr_kernel<4>, rg.grid, rg.block, 0, as_stream(stream), src, order, rowptr, m, c, c / 4, out, arg);
  } else {
    RowGrid rg = row_grid(m, c);
    hipLaunchKernelGGL(scatter_max_csr_kernel<1>, rg.grid, rg.block, 0, as_stream(stream), src, order, rowptr, m, c, c, out, arg);
  }
  return check_launch("pcs_scatter_max_fwd");
}

extern "C" int pcs_scatter_max_bwd_f32(const float *gout, const int32_t *arg, int64_t m, int64_t n, int32_t c,
                                       float *gsrc, void *stream) {
  if (m < 0 || n < 0 || c <= 0) { set_error("pcs_scatter_max_bwd: bad sizes"); return PCS_EINVAL; }
  hipStream_t st = as_stream(stream);
  if (n > 0) {
    if (!gsrc) { set_error("pcs_scatter_max_bwd: null gsrc"); return PCS_EINVAL; }
    if (hipMemsetAsync(gsrc, 0, (size_t)n * c * 4, st) != hipSuccess) { set_error("pcs_scatter_max_bwd: memset failed"); return PCS_ELAUNCH; }
  }
  if (m == 0 || n == 0) return PCS_OK;
  if (!gout || !arg) { set_error("pcs_scatter_max_bwd: null pointer"); return PCS_EINVAL; }
  RowGrid rg = row_grid(m, c);
  hipLaunchKernelGGL(scatter_max_bwd_kernel, rg.grid, rg.block, 0, st, gout, arg, m, c, gsrc);
  return check_launch("pcs_scatter_max_bwd");
}

extern "C" int pcs_map_count(const int32_t *pxpy, int64_t n, int32_t B, int32_t H, int32_t W, int32_t *out,
                             void *stream) {
  if (n < 0 || B <= 0 || H <= 0 || W <= 0 || !out) { set_error("pcs_map_count: bad args"); return PCS_EINVAL; }
  hipStream_t st = as_stream(stream);
  if (hipMemsetAsync(out, 0, (size_t)B * H * W * 4, st) != hipSuccess) { set_error("pcs_map_count: memset failed"); return PCS_ELAUNCH; }
  if (n == 0) return PCS_OK;
  if (!pxpy) { set_error("pcs_map_count: null pxpy"); return PCS_EINVAL; }
  hipLaunchKernelGGL(map_count_kernel, dim3(stream_grid(n, 256)), dim3(256), 0, st, pxpy, n, B, H, W, out);
  return check_launch("pcs_map_count");
}

extern "C" int pcs_denselize_fwd_f32(const float *feat, const int32_t *count_map, const int32_t *pxpy, int64_t n,
                                     int32_t B, int32_t C, int32_t H, int32_t W, float *out, void *stream) {
  if (n < 0 || B <= 0 || C <= 0 || H <= 0 || W <= 0 || !out || !count_map) { set_error("pcs_denselize_fwd: bad args"); return PCS_EINVAL; }
  hipStream_t st = as_stream(stream);
  if (hipMemsetAsync(out, 0, (size_t)B * C * H * W * 4, st) != hipSuccess) { set_error("pcs_denselize_fwd: memset failed"); return PCS_ELAUNCH; }
  if (n == 0) return PCS_OK;
  if (!feat || !pxpy) { set_error("pcs_denselize_fwd: null pointer"); return PCS_EINVAL; }
  RowGrid rg = row_grid(n, C);
  hipLaunchKernelGGL(denselize_fwd_kernel, rg.grid, rg.block, 0, st, feat, count_map, pxpy, n, B, C, H, W, out);
  return check_launch("pcs_denselize_fwd");
}

static int denselize_csr(bool bwd, const float *feat, float *gfeat, const int64_t *order, const int64_t *rowptr,
                         const int32_t *count_map, int32_t B, int32_t C, int32_t H, int32_t W, float *img, hipStream_t st) {
  const int64_t blocks = (int64_t)B * H * ceil_div(W, DN_PX);
  if (blocks > 0x7FFFFFFF) { set_error("pcs_denselize_csr: grid too large"); return PCS_EUNSUPPORTED; }
  if (C <= 32) {
    dim3 grid((unsigned)blocks, 1);
    if (bwd) hipLaunchKernelGGL((denselize_csr_kernel<32, true>), grid, dim3(256), 0, st, feat, gfeat, order, rowptr, count_map, B, C, H, W, img);
    else hipLaunchKernelGGL((denselize_csr_kernel<32, false>), grid, dim3(256), 0, st, feat, gfeat, order, rowptr, count_map, B, C, H, W, img);
  } else {
    dim3 grid((unsigned)blocks, (unsigned)ceil_div(C, 64));
    if (bwd) hipLaunchKernelGGL((denselize_csr_kernel<64, true>), grid, dim3(256), 0, st, feat, gfeat, order, rowptr, count_map, B, C, H, W, img);
    else hipLaunchKernelGGL((denselize_csr_kernel<64, false>), grid, dim3(256), 0, st, feat, gfeat, order, rowptr, count_map, B, C, H, W, img);
  }
  return check_launch("pcs_denselize_csr");
}

extern "C" int pcs_denselize_fwd_csr_f32(const float *feat, const int64_t *order, const int64_t *rowptr,
                                         const int32_t *count_map, int32_t B, int32_t C, int32_t H, int32_t W,
                                         float *out, void *stream) {
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || !out || !count_map || !rowptr) { set_error("pcs_denselize_fwd_csr: bad args"); return PCS_EINVAL; }
  if ((C & 3) || ((uintptr_t)feat & 15)) { set_error("pcs_denselize_fwd_csr: needs C % 4 == 0 and 16-byte aligned rows"); return PCS_EUNSUPPORTED; }
  return denselize_csr(false, feat, nullptr, order, rowptr, count_map, B, C, H, W, out, as_stream(stream));
}

extern "C" int pcs_denselize_bwd_csr_f32(const float *gout, const int64_t *order, const int64_t *rowptr,
                                         const int32_t *count_map, int64_t n, int32_t B, int32_t C, int32_t H, int32_t W,
                                         float *gfeat, void *stream) {
  if (n < 0 || B <= 0 || C <= 0 || H <= 0 || W <= 0 || !count_map || !rowptr) { set_error("pcs_denselize_bwd_csr: bad args"); return PCS_EINVAL; }
  if (n == 0) return PCS_OK;
  if (!gout || !gfeat || !order) { set_error("pcs_denselize_bwd_csr: null pointer"); return PCS_EINVAL; }
  if ((C & 3) || ((uintptr_t)gfeat & 15)) { set_error("pcs_denselize_bwd_csr: needs C % 4 == 0 and 16-byte aligned rows"); return PCS_EUNSUPPORTED; }
  hipStream_t st = as_stream(stream);
  if (hipMemsetAsync(gfeat, 0, (size_t)n * C * 4, st) != hipSuccess) { set_error("pcs_denselize_bwd_csr: memset failed"); return PCS_ELAUNCH; }
  return denselize_csr(true, nullptr, gfeat, order, rowptr, count_map, B, C, H, W, const_cast<float *>(gout), st);
}

extern "C" int pcs_denselize_bwd_f32(const float *gout, const int32_t *count_map, const int32_t *pxpy, int64_t n,
                                     int32_t B, int32_t C, int32_t H, int32_t W, float *gfeat, void *stream) {
  if (n < 0 || B <= 0 || C <= 0 || H <= 0 || W <= 0) { set_error("pcs_denselize_bwd: bad sizes"); return PCS_EINVAL; }
  if (n == 0) return PCS_OK;
  if (!gout || !count_map || !pxpy || !gfeat) { set_error("pcs_denselize_bwd: null pointer"); return PCS_EINVAL; }
  RowGrid rg = row_grid(n, C);
  hipLaunchKernelGGL(denselize_bwd_kernel, rg.grid, rg.block, 0, as_stream(stream), gout, count_map, pxpy, n, B, C, H, W, gfeat);
  return check_launch("pcs_denselize_bwd");
}
