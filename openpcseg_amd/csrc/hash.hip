// K1/K2 coordinate hashing, K3-K5 hash table build/query, K6 count -- gfx950.
// Reference semantics: TS:torchsparse/backend/hash/hash_cuda.cu, hashmap/hashmap_cuda.cu,
// others/query_cuda.cu, others/count_cuda.cu (cited per function in include/pcseg_hip.h).
// All of these are HBM/L2-latency bound integer kernels: one 16-byte coordinate row per lane
// (int4 load), 8-byte stores, grid-stride over >= 2048 workgroups.
#include <stdarg.h>
#include <string.h>

#include "pcs_common.h"

namespace pcs {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace pcs

using namespace pcs;

extern "C" int pcs_abi_version(void) { return PCS_ABI_VERSION; }
extern "C" const char *pcs_last_error(void) { return pcs::g_err; }

// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) hash_kernel(const int4 *__restrict__ coords, int64_t n,
                                                   int64_t *__restrict__ out) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    int4 c = coords[i];
    out[i] = (int64_t)fnv60(c.x, c.y, c.z, c.w);
  }
}

extern "C" int pcs_hash(const int32_t *coords, int64_t n, int64_t *out, void *stream) {
  if (n < 0 || (n > 0 && (!coords || !out))) { set_error("pcs_hash: bad args"); return PCS_EINVAL; }
  if (((uintptr_t)coords & 15) != 0) { set_error("pcs_hash: coords must be 16-byte aligned"); return PCS_EINVAL; }
  if (n == 0) return PCS_OK;
  hipLaunchKernelGGL(hash_kernel, dim3(stream_grid(n, 256)), dim3(256), 0, as_stream(stream),
                     reinterpret_cast<const int4 *>(coords), n, out);
  return check_launch("pcs_hash");
}

// out is (K, n) k-major: blockIdx.y = k keeps both the coord loads and the stores coalesced
__global__ void __launch_bounds__(256) kernel_hash_kernel(const int4 *__restrict__ coords,
                                                          int64_t n,
                                                          const int32_t *__restrict__ offsets,
                                                          int64_t *__restrict__ out) {
  const int k = blockIdx.y;
  const int ox = offsets[3 * k + 0], oy = offsets[3 * k + 1], oz = offsets[3 * k + 2];
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t *o = out + (int64_t)k * n;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    int4 c = coords[i];
    o[i] = (int64_t)fnv60(c.x + ox, c.y + oy, c.z + oz, c.w);
  }
}

extern "C" int pcs_kernel_hash(const int32_t *coords, int64_t n, const int32_t *offsets,
                               int32_t K, int64_t *out, void *stream) {
  if (n < 0 || K < 0 || K > 65535) { set_error("pcs_kernel_hash: bad sizes"); return PCS_EINVAL; }
  if (n == 0 || K == 0) return PCS_OK;
  if (!coords || !offsets || !out || ((uintptr_t)coords & 15)) { set_error("pcs_kernel_hash: bad pointers"); return PCS_EINVAL; }
  int gx = stream_grid(n, 256);
  if (gx > 1024) gx = 1024;
  hipLaunchKernelGGL(kernel_hash_kernel, dim3(gx, K), dim3(256), 0, as_stream(stream),
                     reinterpret_cast<const int4 *>(coords), n, offsets, out);
  return check_launch("pcs_kernel_hash");
}

// ------------------------------------------------------------------------------------------
extern "C" int64_t pcs_hashtable_capacity(int64_t n) {
  int64_t cap = 1024;
  while (cap < 2 * n) cap <<= 1;  // load factor <= 0.5
  return cap;
}
extern "C" size_t pcs_hashtable_bytes(int64_t capacity) {
  return (size_t)capacity * (sizeof(uint64_t) + sizeof(int32_t));
}

__global__ void __launch_bounds__(256) table_insert_kernel(const int64_t *__restrict__ keys,
                                                           int64_t n, uint64_t *tkeys,
                                                           int32_t *tvals, uint64_t mask) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const uint64_t key = (uint64_t)keys[i];
    uint64_t s = slot_of(key, mask);
    while (true) {
      unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long *>(&tkeys[s]),
                                         (unsigned long long)kEmptyKey, (unsigned long long)key);
      if (old == kEmptyKey || old == key) {
        atomicMin(&tvals[s], (int32_t)i);  // duplicate keys: smallest position wins
        break;
      }
      s = (s + 1) & mask;
    }
  }
}

extern "C" int pcs_hashtable_build(const int64_t *keys, int64_t n, void *table,
                                   int64_t capacity, void *stream) {
  if (n < 0 || n >= 0x7F000000LL || capacity < 2 * n || capacity <= 0 ||
      (capacity & (capacity - 1)) || !table || (n > 0 && !keys)) {
    set_error("pcs_hashtable_build: bad args (n=%lld capacity=%lld)", (long long)n, (long long)capacity);
    return PCS_EINVAL;
  }
  uint64_t *tkeys = reinterpret_cast<uint64_t *>(table);
  int32_t *tvals = reinterpret_cast<int32_t *>(tkeys + capacity);
  hipStream_t st = as_stream(stream);
  if (hipMemsetAsync(tkeys, 0xFF, (size_t)capacity * 8, st) != hipSuccess ||
      hipMemsetAsync(tvals, 0x7F, (size_t)capacity * 4, st) != hipSuccess) {
    set_error("pcs_hashtable_build: memset failed");
    return PCS_ELAUNCH;
  }
  if (n == 0) return PCS_OK;
  hipLaunchKernelGGL(table_insert_kernel, dim3(stream_grid(n, 256)), dim3(256), 0, st, keys, n,
                     tkeys, tvals, (uint64_t)capacity - 1);
  return check_launch("pcs_hashtable_build");
}

__global__ void __launch_bounds__(256) table_query_kernel(TableView t,
                                                          const int64_t *__restrict__ q,
                                                          int64_t n1, int64_t *__restrict__ out) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n1; i += stride) {
    out[i] = (int64_t)table_lookup(t, (uint64_t)q[i]) + 1;
  }
}

extern "C" int pcs_hashtable_query(const void *table, int64_t capacity, const int64_t *queries,
                                   int64_t n1, int64_t *out, void *stream) {
  if (!table || capacity <= 0 || (capacity & (capacity - 1)) || n1 < 0 ||
      (n1 > 0 && (!queries || !out))) {
    set_error("pcs_hashtable_query: bad args");
    return PCS_EINVAL;
  }
  if (n1 == 0) return PCS_OK;
  hipLaunchKernelGGL(table_query_kernel, dim3(stream_grid(n1, 256)), dim3(256), 0,
                     as_stream(stream), make_view(table, capacity), queries, n1, out);
  return check_launch("pcs_hashtable_query");
}

// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) count_kernel(const int32_t *__restrict__ idx, int64_t n,
                                                    int32_t *out, int64_t s) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    int32_t v = idx[i];
    if (v >= 0 && v < s) atomicAdd(&out[v], 1);
  }
}

extern "C" int pcs_count(const int32_t *idx, int64_t n, int32_t *out, int64_t s, void *stream) {
  if (n < 0 || s < 0 || (s > 0 && !out) || (n > 0 && !idx)) { set_error("pcs_count: bad args"); return PCS_EINVAL; }
  hipStream_t st = as_stream(stream);
  if (s > 0 && hipMemsetAsync(out, 0, (size_t)s * 4, st) != hipSuccess) {
    set_error("pcs_count: memset failed");
    return PCS_ELAUNCH;
  }
  if (n == 0 || s == 0) return PCS_OK;
  hipLaunchKernelGGL(count_kernel, dim3(stream_grid(n, 256)), dim3(256), 0, st, idx, n, out, s);
  return check_launch("pcs_count");
}
