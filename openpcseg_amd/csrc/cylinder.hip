// Cylinder3D front-end on the device -- gfx950, HBM-bound streaming kernels.
// The reference runs this per frame in the dataloader workers with NumPy
// (R:pcseg/data/dataset/semantickitti/semantickitti_cylinder.py: cart2polar :19-22, the partition :144-160,
// voxelize_with_label :31-45) and maps voxel predictions back to points on the host at eval time
// (R:pcseg/model/segmentor/voxel/minkunet/minkunet.py:441-453). For scans already resident in HBM:
//   1. cyl_partition_kernel : (x,y,z,extras) -> polar (rho, phi in degrees, z) in the reference's float32 arithmetic,
//      clip + cell index + cell-centre features in the reference's float64 arithmetic; one pass, 16-byte row loads.
//   2. (caller) pcs_quantize_* on the cell indices = the reference's sparse_quantize(point_coord).
//   3. label_vote_kernel + label_argmax_kernel : per-voxel class histogram (labels == ignore skipped) and its first
//      arg-max -- integer atomics, so the result does not depend on the order of arrival.
//   4. rows_argmax_gather_kernel : point_predict[i] = argmax(logits[inverse_map[i]]), the eval-time mapping.
// Arithmetic notes (bit-parity with NumPy): float32 products / sums / quotients are the _rn intrinsics (no FMA
// contraction); sqrtf is IEEE; arctan2 is evaluated in double and rounded once to float32 -- NumPy calls the host
// libm's atan2f, which is within one ulp of that on glibc < 2.41, so a point whose angle lies within one float32 ulp
// of a cell face may land in the neighbouring cell (tests/test_cylinder_frontend.py bounds exactly that).
#include <math.h>

#include "pcs_common.h"

using namespace pcs;

namespace {

struct CylCfg {
  double lo[3], hi[3], interval[3];
};

__global__ void __launch_bounds__(256) cyl_partition_kernel(const float *__restrict__ pts, int64_t n, int stride, CylCfg cfg,
                                                            float *__restrict__ polar, int32_t *__restrict__ coord,
                                                            float *__restrict__ feat) {
  const int extras = stride - 3, fdim = 8 + extras;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float *p = pts + i * stride;
    const float x = p[0], y = p[1], z = p[2];
    // cart2polar (:19-22), float32 like NumPy on a float32 scan
    const float rho = sqrtf(__fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y)));
    const float phi = (float)atan2((double)y, (double)x);
    // xyz_pol[:, 1] / np.pi * 180. (:145): float32 array with Python scalars -> float32 arithmetic
    const float deg = __fmul_rn(__fdiv_rn(phi, 3.14159265358979323846f), 180.0f);
    const float pol[3] = {rho, deg, z};
    int idx[3];
    float centre[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      // np.clip(float32, int64 bounds) -> float64; (c - min) / intervals; floor; astype(int) (:153)
      double c = (double)pol[d];
      c = c < cfg.lo[d] ? cfg.lo[d] : (c > cfg.hi[d] ? cfg.hi[d] : c);
      idx[d] = (int)floor((c - cfg.lo[d]) / cfg.interval[d]);
      // (point_coord.astype(float32) + 0.5) * intervals + min_bound (:157), float64, stored as float32 (:165)
      centre[d] = (float)(((double)((float)idx[d] + 0.5f)) * cfg.interval[d] + cfg.lo[d]);
    }
    if (polar) { polar[i * 3 + 0] = pol[0]; polar[i * 3 + 1] = pol[1]; polar[i * 3 + 2] = pol[2]; }
    coord[i * 3 + 0] = idx[0]; coord[i * 3 + 1] = idx[1]; coord[i * 3 + 2] = idx[2];
    if (feat) {  // [cell centre (3), polar (3), x, y, extras...] (:159)
      float *f = feat + i * fdim;
      f[0] = centre[0]; f[1] = centre[1]; f[2] = centre[2];
      f[3] = pol[0]; f[4] = pol[1]; f[5] = pol[2];
      f[6] = x; f[7] = y;
      for (int e = 0; e < extras; ++e) f[8 + e] = p[3 + e];
    }
  }
}

// counter[v][label] += 1 for every point whose label is not `ignore` (:36-37). bad[0] |= 1 when a counted label is
// outside [0, num_classes) -- where the reference raises IndexError.
__global__ void __launch_bounds__(256) label_vote_kernel(const int64_t *__restrict__ inverse, const int64_t *__restrict__ labels,
                                                         int64_t n, int64_t m, int num_classes, int64_t ignore,
                                                         int32_t *__restrict__ counter, int32_t *__restrict__ bad) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t l = labels[i];
    if (l == ignore) continue;
    const int64_t v = inverse[i];
    if (l < 0 || l >= num_classes || v < 0 || v >= m) { atomicOr(bad, 1); continue; }
    atomicAdd(&counter[v * num_classes + l], 1);
  }
}

// voxel_labels = np.argmax(counter, axis=1) (:38): first maximum
__global__ void __launch_bounds__(256) label_argmax_kernel(const int32_t *__restrict__ counter, int64_t m, int num_classes,
                                                           int64_t *__restrict__ out) {
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < m; v += (int64_t)gridDim.x * blockDim.x) {
    const int32_t *c = counter + v * num_classes;
    int best = 0, bv = c[0];
    for (int j = 1; j < num_classes; ++j)
      if (c[j] > bv) { bv = c[j]; best = j; }
    out[v] = best;
  }
}

// out[i] = argmax_j logits[inv ? inv[i] : i][j] (first maximum, like torch.argmax on a row without ties / NaN);
// 16 lanes per row so that a row of <= 64 classes is one or two coalesced loads per lane group.
__global__ void __launch_bounds__(256) rows_argmax_gather_kernel(const float *__restrict__ logits, int64_t m, int c,
                                                                 const int64_t *__restrict__ inv, int64_t n,
                                                                 int64_t *__restrict__ out) {
  const int sub = threadIdx.x & 15;
  const int64_t rows_per_block = blockDim.x >> 4;
  for (int64_t i = (int64_t)blockIdx.x * rows_per_block + (threadIdx.x >> 4); i < n; i += (int64_t)gridDim.x * rows_per_block) {
    int64_t r = inv ? inv[i] : i;
    const bool ok = r >= 0 && r < m;
    const float *row = logits + (ok ? r : 0) * c;
    float bv = -INFINITY;
    int bj = 0x7FFFFFFF;
    for (int j = sub; j < c; j += 16) {
      const float v = row[j];
      if (v > bv || (v == bv && j < bj)) { bv = v; bj = j; }
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bv, o, 64);
      const int oj = __shfl_xor(bj, o, 64);
      if (ov > bv || (ov == bv && oj < bj)) { bv = ov; bj = oj; }
    }
    if (sub == 0) out[i] = ok ? (bj == 0x7FFFFFFF ? 0 : bj) : -1;
  }
}

}  // namespace

extern "C" int pcs_cylinder_partition_f32(const float *points, int64_t n, int32_t row_stride, const double *space_min3,
                                          const double *space_max3, const int32_t *grid3, float *polar, int32_t *coord,
                                          float *feat, void *stream) {
  if (n < 0 || row_stride < 3 || !space_min3 || !space_max3 || !grid3) { set_error("pcs_cylinder_partition_f32: bad args"); return PCS_EINVAL; }
  CylCfg cfg;
  for (int d = 0; d < 3; ++d) {
    if (grid3[d] < 2 || !(space_max3[d] > space_min3[d])) { set_error("pcs_cylinder_partition_f32: grid must be >= 2 and max > min"); return PCS_EINVAL; }
    cfg.lo[d] = space_min3[d];
    cfg.hi[d] = space_max3[d];
    cfg.interval[d] = (space_max3[d] - space_min3[d]) / (double)(grid3[d] - 1);  // crop_range / (grid - 1) (:151)
  }
  if (n == 0) return PCS_OK;
  if (!points || !coord) { set_error("pcs_cylinder_partition_f32: null pointer"); return PCS_EINVAL; }
  hipLaunchKernelGGL(cyl_partition_kernel, dim3(stream_grid(n, 256)), dim3(256), 0, as_stream(stream), points, n, row_stride,
                     cfg, polar, coord, feat);
  return check_launch("pcs_cylinder_partition_f32");
}

extern "C" int pcs_voxel_label_vote(const int64_t *inverse, const int64_t *labels, int64_t n, int64_t m, int32_t num_classes,
                                    int64_t ignore_label, int32_t *counter_ws, int32_t *bad_flag, int64_t *voxel_labels,
                                    void *stream) {
  if (n < 0 || m < 0 || num_classes <= 0) { set_error("pcs_voxel_label_vote: bad sizes"); return PCS_EINVAL; }
  if (m == 0) return PCS_OK;
  if (!counter_ws || !bad_flag || !voxel_labels || (n > 0 && (!inverse || !labels))) { set_error("pcs_voxel_label_vote: null pointer"); return PCS_EINVAL; }
  hipStream_t st = as_stream(stream);
  if (hipMemsetAsync(counter_ws, 0, (size_t)m * num_classes * sizeof(int32_t), st) != hipSuccess ||
      hipMemsetAsync(bad_flag, 0, sizeof(int32_t), st) != hipSuccess) {
    set_error("pcs_voxel_label_vote: memset failed");
    return PCS_ELAUNCH;
  }
  if (n > 0)
    hipLaunchKernelGGL(label_vote_kernel, dim3(stream_grid(n, 256)), dim3(256), 0, st, inverse, labels, n, m, num_classes,
                       ignore_label, counter_ws, bad_flag);
  hipLaunchKernelGGL(label_argmax_kernel, dim3(stream_grid(m, 256)), dim3(256), 0, st, counter_ws, m, num_classes, voxel_labels);
  return check_launch("pcs_voxel_label_vote");
}

extern "C" int pcs_rows_argmax_gather_f32(const float *logits, int64_t m, int32_t c, const int64_t *inverse, int64_t n,
                                          int64_t *out, void *stream) {
  if (n < 0 || m < 0 || c <= 0) { set_error("pcs_rows_argmax_gather_f32: bad sizes"); return PCS_EINVAL; }
  if (n == 0) return PCS_OK;
  if (!logits || !out || m == 0) { set_error("pcs_rows_argmax_gather_f32: null pointer / empty logits"); return PCS_EINVAL; }
  hipLaunchKernelGGL(rows_argmax_gather_kernel, dim3(stream_grid(n * 16, 256)), dim3(256), 0, as_stream(stream), logits, m, c,
                     inverse, n, out);
  return check_launch("pcs_rows_argmax_gather_f32");
}
