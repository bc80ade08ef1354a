// Shared helpers for the gfx950 kernels behind include/pcseg_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/pcseg_hip.h"

namespace pcs {

void set_error(const char *fmt, ...);

// wave width on CDNA4; hard-coded on purpose (guide: never assume 32)
constexpr int kWave = 64;

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

inline int check_launch(const char *what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return PCS_ELAUNCH;
  }
  return PCS_OK;
}

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// grid for a memory-bound grid-stride kernel: enough workgroups to fill 256 CUs x 8
inline int stream_grid(int64_t work_items, int block) {
  int64_t g = ceil_div(work_items, block);
  if (g < 1) g = 1;
  if (g > 256 * 8) g = 256 * 8;
  return (int)g;
}

// ---- 60-bit FNV-1a of a coordinate row (reference: hash_cuda.cu:10-23) -------------------
__device__ __forceinline__ uint64_t fnv60(int x, int y, int z, int b) {
  uint64_t h = 14695981039346656037ULL;
  h ^= (uint32_t)x; h *= 1099511628211ULL;
  h ^= (uint32_t)y; h *= 1099511628211ULL;
  h ^= (uint32_t)z; h *= 1099511628211ULL;
  h ^= (uint32_t)b; h *= 1099511628211ULL;
  return (h >> 60) ^ (h & 0x0FFFFFFFFFFFFFFFULL);
}

// ---- open-addressing table ---------------------------------------------------------------
constexpr uint64_t kEmptyKey = 0xFFFFFFFFFFFFFFFFULL;  // hashes are < 2^60, never this

__device__ __forceinline__ uint64_t slot_of(uint64_t key, uint64_t mask) {
  // the key is itself an FNV hash, but its low bits come straight from the last multiply;
  // one more multiplicative mix spreads neighbouring voxels over the table
  uint64_t m = key * 0x9E3779B97F4A7C15ULL;
  return (m >> 20) & mask;
}

struct TableView {
  const uint64_t *keys;
  const int32_t *vals;
  uint64_t mask;
};

__device__ __forceinline__ int32_t table_lookup(const TableView &t, uint64_t key) {
  uint64_t s = slot_of(key, t.mask);
  while (true) {
    uint64_t k = t.keys[s];
    if (k == key) return t.vals[s];
    if (k == kEmptyKey) return -1;
    s = (s + 1) & t.mask;
  }
}

inline TableView make_view(const void *table, int64_t capacity) {
  TableView v;
  v.keys = reinterpret_cast<const uint64_t *>(table);
  v.vals = reinterpret_cast<const int32_t *>(v.keys + capacity);
  v.mask = (uint64_t)capacity - 1;
  return v;
}

}  // namespace pcs
