// Device-side voxel dedup of one scan: the reference runs it in the dataloader workers
// with NumPy (TS:torchsparse/utils/quantize.py:9-46: floor(coords / voxel_size) -> ravel_hash -> np.unique with
// return_index / return_inverse). Same contract here -- one representative per voxel = FIRST occurrence, voxels
// ordered by ascending ravel hash -- as four streaming passes around one stable radix sort of (key, row):
//   1. quantize_floor_kernel : floor(double(p) / double(v)) -> int32 coords, bounding box by wave-reduced atomics
//   2. quantize_key_kernel   : row-major linear index inside the bounding box (the reference's ravel_hash)
//   3. (caller) stable sort of the keys with the row index as payload
//   4. quantize_flag_kernel  : 1 at the first row of every run of equal keys; (caller) inclusive scan = voxel id + 1
//   5. quantize_emit_kernel  : voxel coords + representative row (run head) + inverse map (row -> voxel)
// All HBM-bound: 12-28 B per point per pass, 16-byte row loads where the layout allows.
#include "pcs_common.h"

using namespace pcs;

namespace {

__device__ __forceinline__ int wave_min(int v) {
  for (int o = 32; o > 0; o >>= 1) { const int t = __shfl_xor(v, o, 64); v = t < v ? t : v; }
  return v;
}
__device__ __forceinline__ int wave_max(int v) {
  for (int o = 32; o > 0; o >>= 1) { const int t = __shfl_xor(v, o, 64); v = t > v ? t : v; }
  return v;
}

// bbox[0..2] = min, bbox[3..5] = max (initialised to INT_MAX / INT_MIN by the caller)
template <typename TIn>
__global__ void __launch_bounds__(256) quantize_floor_kernel(const TIn *__restrict__ pts, int64_t n, int stride,
                                                             double vx, double vy, double vz,
                                                             int32_t *__restrict__ coords, int32_t *__restrict__ bbox) {
  int mn[3] = {INT32_MAX, INT32_MAX, INT32_MAX}, mx[3] = {INT32_MIN, INT32_MIN, INT32_MIN};
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const TIn *p = pts + i * stride;
    const int c[3] = {(int)floor((double)p[0] / vx), (int)floor((double)p[1] / vy), (int)floor((double)p[2] / vz)};
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      coords[i * 3 + d] = c[d];
      mn[d] = c[d] < mn[d] ? c[d] : mn[d];
      mx[d] = c[d] > mx[d] ? c[d] : mx[d];
    }
  }
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const int a = wave_min(mn[d]), b = wave_max(mx[d]);
    if ((threadIdx.x & 63) == 0) { atomicMin(&bbox[d], a); atomicMax(&bbox[3 + d], b); }
  }
}

__global__ void __launch_bounds__(256) quantize_key_kernel(const int32_t *__restrict__ coords, int64_t n,
                                                           const int32_t *__restrict__ bbox, int64_t *__restrict__ keys) {
  const int64_t x0 = bbox[0], y0 = bbox[1], z0 = bbox[2];
  const int64_t ey = (int64_t)bbox[4] - y0 + 1, ez = (int64_t)bbox[5] - z0 + 1;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t x = coords[i * 3 + 0] - x0, y = coords[i * 3 + 1] - y0, z = coords[i * 3 + 2] - z0;
    keys[i] = (x * ey + y) * ez + z;  // ((0 + x) * ey + y) * ez + z, quantize.py:15-21
  }
}

// whole batch at once (hostdata.sparse_quantize_frames): frame on top of the BATCH's bounding box -- ascending key = frames in order,
// inside a frame ascending (x, y, z) = the reference's per-frame ravel hash order. frames: int64 per row.
__global__ void __launch_bounds__(256) quantize_frame_key_kernel(const int32_t *__restrict__ coords, const int64_t *__restrict__ frames,
                                                                 int64_t n, const int32_t *__restrict__ bbox, int64_t *__restrict__ keys) {
  const int64_t x0 = bbox[0], y0 = bbox[1], z0 = bbox[2];
  const int64_t ex = (int64_t)bbox[3] - x0 + 1, ey = (int64_t)bbox[4] - y0 + 1, ez = (int64_t)bbox[5] - z0 + 1;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t x = coords[i * 3 + 0] - x0, y = coords[i * 3 + 1] - y0, z = coords[i * 3 + 2] - z0;
    keys[i] = ((frames[i] * ex + x) * ey + y) * ez + z;
  }
}

__global__ void __launch_bounds__(256) quantize_flag_kernel(const int64_t *__restrict__ sorted_keys, int64_t n,
                                                            int32_t *__restrict__ flags) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    flags[i] = (i == 0 || sorted_keys[i] != sorted_keys[i - 1]) ? 1 : 0;
}

// rank[i] = inclusive scan of the flags = 1-based voxel of sorted position i; a stable sort puts the smallest
// row of a voxel at its run head
__global__ void __launch_bounds__(256) quantize_emit_kernel(const int32_t *__restrict__ flags, const int64_t *__restrict__ rank,
                                                            const int64_t *__restrict__ perm, const int32_t *__restrict__ coords,
                                                            int64_t n, int32_t *__restrict__ vox, int64_t *__restrict__ index,
                                                            int64_t *__restrict__ inverse) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = perm[i], v = rank[i] - 1;
    if (inverse) inverse[row] = v;
    if (flags[i]) {
      if (index) index[v] = row;
      vox[v * 3 + 0] = coords[row * 3 + 0];
      vox[v * 3 + 1] = coords[row * 3 + 1];
      vox[v * 3 + 2] = coords[row * 3 + 2];
    }
  }
}

// Run heads of a sorted key vector -> unique keys, inverse map and CSR row pointers in one pass (the voxel set of
// initial_voxelize: R:pcseg/model/segmentor/voxel/minkunet/utils.py:16-19 does torch.unique + a hash-table query + a
// histogram for the same three results). rank = inclusive scan of the flags.
__global__ void __launch_bounds__(256) unique_emit_kernel(const int32_t *__restrict__ flags, const int64_t *__restrict__ rank,
                                                          const int64_t *__restrict__ perm, const int64_t *__restrict__ skeys,
                                                          int64_t n, int64_t *__restrict__ uniq, int64_t *__restrict__ inverse,
                                                          int64_t *__restrict__ rowptr) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t v = rank[i] - 1;
    inverse[perm[i]] = v;
    if (flags[i]) { uniq[v] = skeys[i]; rowptr[v] = i; }
    if (i == n - 1) rowptr[v + 1] = n;
  }
}

}  // namespace

extern "C" int pcs_unique_emit(const int32_t *flags, const int64_t *rank, const int64_t *perm, const int64_t *sorted_keys,
                               int64_t n, int64_t *uniq, int64_t *inverse, int64_t *rowptr, void *stream) {
  if (n < 0) { set_error("pcs_unique_emit: bad size"); return PCS_EINVAL; }
  if (n == 0) return PCS_OK;
  if (!flags || !rank || !perm || !sorted_keys || !uniq || !inverse || !rowptr) { set_error("pcs_unique_emit: null pointer"); return PCS_EINVAL; }
  hipLaunchKernelGGL(unique_emit_kernel, dim3(stream_grid(n, 256)), dim3(256), 0, as_stream(stream), flags, rank, perm,
                     sorted_keys, n, uniq, inverse, rowptr);
  return check_launch("pcs_unique_emit");
}

extern "C" int pcs_quantize_floor(const void *points, int32_t is_float, int64_t n, int32_t row_stride,
                                  const double *voxel_size3, int32_t *coords, int32_t *bbox, void *stream) {
  if (n < 0 || row_stride < 3 || is_float < 0 || is_float > 2 || !voxel_size3 || !(voxel_size3[0] > 0) || !(voxel_size3[1] > 0) || !(voxel_size3[2] > 0)) {
    set_error("pcs_quantize_floor: bad args");
    return PCS_EINVAL;
  }
  if (n == 0) return PCS_OK;
  if (!points || !coords || !bbox) { set_error("pcs_quantize_floor: null pointer"); return PCS_EINVAL; }
  hipStream_t st = as_stream(stream);
  const int g = stream_grid(n, 256);
  if (is_float == 2)
    hipLaunchKernelGGL(quantize_floor_kernel<double>, dim3(g), dim3(256), 0, st, reinterpret_cast<const double *>(points), n,
                       row_stride, voxel_size3[0], voxel_size3[1], voxel_size3[2], coords, bbox);
  else if (is_float)
    hipLaunchKernelGGL(quantize_floor_kernel<float>, dim3(g), dim3(256), 0, st, reinterpret_cast<const float *>(points), n,
                       row_stride, voxel_size3[0], voxel_size3[1], voxel_size3[2], coords, bbox);
  else
    hipLaunchKernelGGL(quantize_floor_kernel<int32_t>, dim3(g), dim3(256), 0, st, reinterpret_cast<const int32_t *>(points), n,
                       row_stride, voxel_size3[0], voxel_size3[1], voxel_size3[2], coords, bbox);
  return check_launch("pcs_quantize_floor");
}

extern "C" int pcs_quantize_keys(const int32_t *coords, int64_t n, const int32_t *bbox, int64_t *keys, void *stream) {
  if (n < 0) { set_error("pcs_quantize_keys: bad size"); return PCS_EINVAL; }
  if (n == 0) return PCS_OK;
  if (!coords || !bbox || !keys) { set_error("pcs_quantize_keys: null pointer"); return PCS_EINVAL; }
  hipLaunchKernelGGL(quantize_key_kernel, dim3(stream_grid(n, 256)), dim3(256), 0, as_stream(stream), coords, n, bbox, keys);
  return check_launch("pcs_quantize_keys");
}

extern "C" int pcs_quantize_frame_keys(const int32_t *coords, const int64_t *frames, int64_t n, const int32_t *bbox, int64_t *keys,
                                       void *stream) {
  if (n < 0) { set_error("pcs_quantize_frame_keys: bad size"); return PCS_EINVAL; }
  if (n == 0) return PCS_OK;
  if (!coords || !frames || !bbox || !keys) { set_error("pcs_quantize_frame_keys: null pointer"); return PCS_EINVAL; }
  hipLaunchKernelGGL(quantize_frame_key_kernel, dim3(stream_grid(n, 256)), dim3(256), 0, as_stream(stream), coords, frames, n, bbox, keys);
  return check_launch("pcs_quantize_frame_keys");
}

extern "C" int pcs_quantize_flags(const int64_t *sorted_keys, int64_t n, int32_t *flags, void *stream) {
  if (n < 0) { set_error("pcs_quantize_flags: bad size"); return PCS_EINVAL; }
  if (n == 0) return PCS_OK;
  if (!sorted_keys || !flags) { set_error("pcs_quantize_flags: null pointer"); return PCS_EINVAL; }
  hipLaunchKernelGGL(quantize_flag_kernel, dim3(stream_grid(n, 256)), dim3(256), 0, as_stream(stream), sorted_keys, n, flags);
  return check_launch("pcs_quantize_flags");
}

extern "C" int pcs_quantize_emit(const int32_t *flags, const int64_t *rank, const int64_t *perm, const int32_t *coords,
                                 int64_t n, int32_t *vox, int64_t *index, int64_t *inverse, void *stream) {
  if (n < 0) { set_error("pcs_quantize_emit: bad size"); return PCS_EINVAL; }
  if (n == 0) return PCS_OK;
  if (!flags || !rank || !perm || !coords || !vox) { set_error("pcs_quantize_emit: null pointer"); return PCS_EINVAL; }
  hipLaunchKernelGGL(quantize_emit_kernel, dim3(stream_grid(n, 256)), dim3(256), 0, as_stream(stream), flags, rank, perm,
                     coords, n, vox, index, inverse);
  return check_launch("pcs_quantize_emit");
}
