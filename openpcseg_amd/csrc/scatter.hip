// Cylindrical scatter (torch_scatter.scatter_max / scatter_mean as used by the reference's cylinder front-end,
// R:tools/utils/common/seg_utils.py:172-188, R:pcseg/model/segmentor/voxel/cylinder3d/cylinder_ts.py:24-43) and the
// range-image scatter of RPVNet (K13-K15, RL:range_utils/src/map_count_gpu.cu:5-14, denselize_gpu.cu:5-34) -- gfx950.
// All HBM-bound. scatter_max runs as a per-voxel segmented reduction over a sorted CSR (deterministic, no float
// atomic-max tricks, argmax = first point in CSR order); the range scatter keeps the reference's atomic dataflow
// (NCHW output, one pixel per point) with the bounds checks the reference lacks.
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
// empty voxels: out = 0, arg = -1.
__global__ void __launch_bounds__(256) scatter_max_csr_kernel(const float *__restrict__ src,
                                                              const int64_t *__restrict__ order,
                                                              const int64_t *__restrict__ rowptr, int64_t m,
                                                              int c, float *__restrict__ out,
                                                              int64_t *__restrict__ arg) {
  for (int64_t v = (int64_t)blockIdx.x * blockDim.y + threadIdx.y; v < m; v += (int64_t)gridDim.x * blockDim.y) {
    const int64_t e0 = rowptr[v], e1 = rowptr[v + 1];
    for (int j = threadIdx.x; j < c; j += blockDim.x) {
      float best = 0.f;
      int64_t bi = -1;
      for (int64_t e = e0; e < e1; ++e) {
        const int64_t i = order[e];
        const float x = src[i * c + j];
        if (bi < 0 || x > best) { best = x; bi = i; }
      }
      out[v * c + j] = best;
      arg[v * c + j] = bi;
    }
  }
}

// grad_src is zero-filled by the caller of the C entry point; every (v, j) routes to exactly one element
__global__ void __launch_bounds__(256) scatter_max_bwd_kernel(const float *__restrict__ gout,
                                                              const int64_t *__restrict__ arg, int64_t m, int c,
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
                                       int32_t c, float *out, int64_t *arg, void *stream) {
  if (m < 0 || c <= 0) { set_error("pcs_scatter_max_fwd: bad sizes"); return PCS_EINVAL; }
  if (m == 0) return PCS_OK;
  if (!src || !order || !rowptr || !out || !arg) { set_error("pcs_scatter_max_fwd: null pointer"); return PCS_EINVAL; }
  RowGrid rg = row_grid(m, c);
  hipLaunchKernelGGL(scatter_max_csr_kernel, rg.grid, rg.block, 0, as_stream(stream), src, order, rowptr, m, c, out, arg);
  return check_launch("pcs_scatter_max_fwd");
}

extern "C" int pcs_scatter_max_bwd_f32(const float *gout, const int64_t *arg, int64_t m, int64_t n, int32_t c,
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

extern "C" int pcs_denselize_bwd_f32(const float *gout, const int32_t *count_map, const int32_t *pxpy, int64_t n,
                                     int32_t B, int32_t C, int32_t H, int32_t W, float *gfeat, void *stream) {
  if (n < 0 || B <= 0 || C <= 0 || H <= 0 || W <= 0) { set_error("pcs_denselize_bwd: bad sizes"); return PCS_EINVAL; }
  if (n == 0) return PCS_OK;
  if (!gout || !count_map || !pxpy || !gfeat) { set_error("pcs_denselize_bwd: null pointer"); return PCS_EINVAL; }
  RowGrid rg = row_grid(n, C);
  hipLaunchKernelGGL(denselize_bwd_kernel, rg.grid, rg.block, 0, as_stream(stream), gout, count_map, pxpy, n, B, C, H, W, gfeat);
  return check_launch("pcs_denselize_bwd");
}
