// Sorted unique of packed coordinate keys: step 2 of spdownsample (TS:torchsparse/nn/functional/downsample.py:47-51, the reference's
// `torch.unique(coords, dim=0)` over [b, x, y, z] rows = ascending order of the packed keys of pcs_downsample_pack) behind the C ABI,
// so that a host without torch can run the whole of a5 (SURVEY.md section 7 "(ii) own sort"). One radix sort (rocPRIM onesweep, the
// device library of this image -- the same sort torch.unique ends up in), one adjacent-unique compaction, and a one-thread kernel that
// leaves everything the host has to read in ONE 24-byte record: the number of unique keys, the largest of them (the general
// branch's "rejected candidate" sentinel sorts last) and the packing error flag. No allocation: workspace from the caller.
//
// pcs_index_csr_i32: the CSR (entries sorted by target row + row pointers) the contention-free forms of K7 / K10 run over
// (pcs_voxelize_fwd_csr_f32, pcs_devoxelize_bwd_csr_f32), from ONE stable radix sort over only the bits a row index below m
// needs (16-21 of 32 on the bench levels: 2-3 digit passes instead of the 4 of a full int32 sort) + a binary search per row.
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_select.hpp>
#include <rocprim/iterator/counting_iterator.hpp>

#include "pcs_common.h"

using namespace pcs;

namespace {

inline size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

struct Plan {
  size_t off_sorted, off_temp, temp_bytes, bytes;
};

int make_plan(int64_t n, Plan &p) {
  p.off_sorted = 0;
  p.off_temp = align_up(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
  size_t t_sort = 0, t_uniq = 0;
  if (n > 0) {
    hipError_t e = rocprim::radix_sort_keys(nullptr, t_sort, (const int64_t *)nullptr, (int64_t *)nullptr, (size_t)n, 0u, 64u, (hipStream_t)0);
    if (e == hipSuccess)
      e = rocprim::unique(nullptr, t_uniq, (const int64_t *)nullptr, (int64_t *)nullptr, (int64_t *)nullptr, (size_t)n,
                          rocprim::equal_to<int64_t>(), (hipStream_t)0);
    if (e != hipSuccess) {
      set_error("pcs_sort_unique: rocprim temporary-storage query failed: %s", hipGetErrorString(e));
      return PCS_ELAUNCH;
    }
  }
  p.temp_bytes = t_sort > t_uniq ? t_sort : t_uniq;
  p.bytes = align_up(p.off_temp + p.temp_bytes) + 256;
  return PCS_OK;
}

__global__ void su_finish_kernel(const int64_t *__restrict__ out, int64_t *__restrict__ info, const int32_t *__restrict__ err) {
  const int64_t m = info[0];
  info[1] = m > 0 ? out[m - 1] : INT64_MIN;
  info[2] = err ? (int64_t)*err : 0;
}

struct CsrPlan {
  size_t off_kin, off_kout, off_temp, temp_bytes, bytes;
  unsigned end_bit;
};

int make_csr_plan(int64_t n, int64_t m, CsrPlan &p) {
  p.end_bit = 1;
  while (p.end_bit < 32 && ((uint64_t)1 << p.end_bit) <= (uint64_t)m) ++p.end_bit;   // keys 0 .. m (m = the "no row" key)
  const size_t kb = align_up(sizeof(uint32_t) * (size_t)(n > 0 ? n : 1));
  p.off_kin = 0; p.off_kout = kb; p.off_temp = 2 * kb;
  p.temp_bytes = 0;
  if (n > 0) {
    hipError_t e = rocprim::radix_sort_pairs(nullptr, p.temp_bytes, (const uint32_t *)nullptr, (uint32_t *)nullptr,
                                             rocprim::counting_iterator<int64_t>(0), (int64_t *)nullptr, (size_t)n, 0u, p.end_bit,
                                             (hipStream_t)0);
    if (e != hipSuccess) {
      set_error("pcs_index_csr: rocprim temporary-storage query failed: %s", hipGetErrorString(e));
      return PCS_ELAUNCH;
    }
  }
  p.bytes = align_up(p.off_temp + p.temp_bytes) + 256;
  return PCS_OK;
}

__global__ void __launch_bounds__(256) csr_keys_kernel(const int32_t *__restrict__ index, int64_t n, int32_t m,
                                                       uint32_t *__restrict__ keys) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int32_t v = index[i];
    keys[i] = (v < 0 || v >= m) ? (uint32_t)m : (uint32_t)v;   // entries without a row sort behind the last row
  }
}

__global__ void __launch_bounds__(256) csr_rowptr_kernel(const uint32_t *__restrict__ sorted, int64_t n, int64_t m,
                                                         int64_t *__restrict__ rowptr) {
  for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v <= m; v += (int64_t)gridDim.x * 256) {
    int64_t lo = 0, hi = n;   // first position whose key is >= v
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if ((int64_t)sorted[mid] < v) lo = mid + 1; else hi = mid;
    }
    rowptr[v] = lo;
  }
}

}  // namespace

extern "C" size_t pcs_index_csr_ws_bytes(int64_t n, int64_t m) {
  CsrPlan p;
  if (n < 0 || m < 0 || m >= 0x7FFFFFFF || make_csr_plan(n, m, p) != PCS_OK) return 0;
  return p.bytes;
}

extern "C" int pcs_index_csr_i32(const int32_t *index, int64_t n, int64_t m, int64_t *order, int64_t *rowptr, void *ws,
                                 size_t ws_bytes, void *stream) {
  if (n < 0 || m < 0 || m >= 0x7FFFFFFF || !rowptr || (n > 0 && (!index || !order))) { set_error("pcs_index_csr_i32: bad args"); return PCS_EINVAL; }
  hipStream_t st = as_stream(stream);
  CsrPlan p;
  int rc = make_csr_plan(n, m, p);
  if (rc) return rc;
  if (n > 0 && (!ws || ws_bytes < p.bytes)) { set_error("pcs_index_csr_i32: workspace too small (pcs_index_csr_ws_bytes)"); return PCS_EWORKSPACE; }
  char *base = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(ws) + 255) & ~(uintptr_t)255);
  uint32_t *kin = reinterpret_cast<uint32_t *>(base + p.off_kin), *kout = reinterpret_cast<uint32_t *>(base + p.off_kout);
  if (n > 0) {
    hipLaunchKernelGGL(csr_keys_kernel, dim3(stream_grid(n, 256)), dim3(256), 0, st, index, n, (int32_t)m, kin);
    size_t tb = p.temp_bytes;
    hipError_t e = rocprim::radix_sort_pairs(base + p.off_temp, tb, (const uint32_t *)kin, kout, rocprim::counting_iterator<int64_t>(0),
                                             order, (size_t)n, 0u, p.end_bit, st);
    if (e != hipSuccess) { set_error("pcs_index_csr_i32: rocprim::radix_sort_pairs: %s", hipGetErrorString(e)); return PCS_ELAUNCH; }
  }
  hipLaunchKernelGGL(csr_rowptr_kernel, dim3(stream_grid(m + 1, 256)), dim3(256), 0, st, kout, n, m, rowptr);
  return check_launch("pcs_index_csr_i32");
}

extern "C" size_t pcs_sort_unique_ws_bytes(int64_t n) {
  Plan p;
  if (n < 0 || make_plan(n, p) != PCS_OK) return 0;
  return p.bytes;
}

extern "C" int pcs_sort_unique_i64(const int64_t *keys, int64_t n, int64_t *out, int64_t *info, const int32_t *err_flag, void *ws,
                                   size_t ws_bytes, void *stream) {
  if (n < 0 || !info || (n > 0 && (!keys || !out))) { set_error("pcs_sort_unique_i64: bad args"); return PCS_EINVAL; }
  hipStream_t st = as_stream(stream);
  if (n == 0) {
    if (hipMemsetAsync(info, 0, sizeof(int64_t), st) != hipSuccess) { set_error("pcs_sort_unique_i64: memset failed"); return PCS_ELAUNCH; }
    hipLaunchKernelGGL(su_finish_kernel, dim3(1), dim3(1), 0, st, out, info, err_flag);
    return check_launch("pcs_sort_unique_i64");
  }
  Plan p;
  int rc = make_plan(n, p);
  if (rc) return rc;
  if (!ws || ws_bytes < p.bytes) { set_error("pcs_sort_unique_i64: workspace too small (pcs_sort_unique_ws_bytes)"); return PCS_EWORKSPACE; }
  char *base = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(ws) + 255) & ~(uintptr_t)255);
  int64_t *sorted = reinterpret_cast<int64_t *>(base + p.off_sorted);
  void *temp = base + p.off_temp;
  size_t tb = p.temp_bytes;
  hipError_t e = rocprim::radix_sort_keys(temp, tb, keys, sorted, (size_t)n, 0u, 64u, st);
  if (e != hipSuccess) { set_error("pcs_sort_unique_i64: rocprim::radix_sort_keys: %s", hipGetErrorString(e)); return PCS_ELAUNCH; }
  tb = p.temp_bytes;
  e = rocprim::unique(temp, tb, (const int64_t *)sorted, out, info, (size_t)n, rocprim::equal_to<int64_t>(), st);
  if (e != hipSuccess) { set_error("pcs_sort_unique_i64: rocprim::unique: %s", hipGetErrorString(e)); return PCS_ELAUNCH; }
  hipLaunchKernelGGL(su_finish_kernel, dim3(1), dim3(1), 0, st, out, info, err_flag);
  return check_launch("pcs_sort_unique_i64");
}
