// Device helpers shared by the ring kernels (conv_ring6h.hip: 16-bit MFMA; conv_ring6f.hip: fp32 MFMA): LDS-DMA, the fence-less
// workgroup barrier, and the launch-shape rules of the fp32 ring kernel (the half kernel's live in conv_half.h).
// NOT part of the product library: compiled only into the variant build `tools/build_variant_lib.sh ring` (-DPCS_WITH_RING=1),
// through openpcseg_amd/csrc/conv_ring.h.
#pragma once
#include "conv_half.h"

namespace pcs {

// ---- the ring kernel's launch shape (conv_ring6h.hip), shared with the tile-height picker and the BatchNorm-partials query ----
#ifndef PCS_RING_D
#define PCS_RING_D 5             /* variant builds (tools/build_variant_lib.sh): ring depth */
#endif
#ifndef PCS_RING_ABLATE
#define PCS_RING_ABLATE 0        /* variant builds: 1 no weight loads, 2 row DMAs of 4 bytes per lane, 3 no commit, 4 no MFMA, 5 no row DMA at all,
                                    6 no DMA completion wait, 7 compute waves pass the barriers only (results are wrong, times tell) */
#endif
constexpr int kRingDepth = PCS_RING_D;  // A ring: batches of gathered rows resident / in flight per workgroup
constexpr int kRingBatchRows = 2;  // 16-row blocks per batch
constexpr int kRingMeta = PCS_RING_D > 9 ? 32 : 16;  // pair-index ring: slots (>= 2 kRingDepth - 2)

struct RingShape {
  int nctt, nc, kc;  // 16-column tiles per column tile, of them per compute wave, 32-channel steps per weight chunk
  int nwaves() const { return nctt / nc + 1; }  // compute waves + the loader
};

// the ring kernel serves cin % 32 == 0 from 64 channels with 2, 3 or 4 steps per chunk (cin = 64, 96, 128, 192, 256, 384, 512 ...)
// and 96 / 128-column tiles (cout = 96, 128, 192, 256, 384 ...)
int &conv_ring_mode();  // conv_ring6h.hip: 0 never, 1 wherever it applies, -1 per-shape policy (pcs_conv_ring_enable / PCS_CONVH_RING)
inline bool conv_ring_policy(int cin, int cout, int K) {
  // where the ring kernel beats conv_os5h_kernel (profiles/round4_ring.md); nowhere yet
  (void)cin; (void)cout; (void)K;
  return false;
}
inline bool conv_ring_shape(int cin, int cout, int K, RingShape *out) {
  static const int force_nc = getenv("PCS_CONVH_RING_NC") ? atoi(getenv("PCS_CONVH_RING_NC")) : 0;
  const int mode = conv_ring_mode();
  if (mode == 0 || !convh_applies(cin, cout, K) || cin % 32) return false;
  if (mode < 0 && !conv_ring_policy(cin, cout, K)) return false;
  const int ns = cin / 32, nctt = conv_nctt(cout);
  if (nctt != 6 && nctt != 8) return false;
  int kc = 0;
  if (ns % 4 == 0) kc = 4;
  else if (ns % 3 == 0) kc = 3;
  else if (ns % 2 == 0) kc = 2;
  if (!kc || ns / kc > 7) return false;
  if (out) { out->nctt = nctt; out->nc = (force_nc == 1 || force_nc == 2) ? force_nc : 2; out->kc = kc; }
  return true;
}
inline int conv_ring_bt_cap(int T, int ns, int kc, int K) {
  const int per_off = (T / 16 + kRingBatchRows - 1) / kRingBatchRows + 1;
  return ((ns / kc) * K * per_off + 15) & ~15;
}
// LDS bytes: [A ring | pair-index ring | offset lists | batch table | accumulator tile]; returns the tile's byte offset in *acc_off
inline size_t conv_ring_lds(int T, const RingShape &s, int ns, int K, int *acc_off) {
  size_t off = (size_t)kRingDepth * kRingBatchRows * s.kc * 1024 + (size_t)kRingMeta * kRingBatchRows * 128;
  off += 4 * 36 * 4 + 16;                                   // kl_k / kl_s / kl_m / kl_b + {nk, NB}
  off = (off + 7) & ~(size_t)7;
  off += (size_t)conv_ring_bt_cap(T, ns, s.kc, K) * 8;      // int2 descriptors
  off = (off + 15) & ~(size_t)15;
  if (acc_off) *acc_off = (int)off;
  return off + (size_t)(T + 1) * (16 * s.nctt + 4) * 4;
}
inline bool conv_ring_applies(int cin, int cout, int K, int T, RingShape *out) {
  RingShape s;
  if (!conv_ring_shape(cin, cout, K, &s)) return false;
  if (T < 32 || T > 512 || conv_ring_lds(T, s, cin / 32, K, nullptr) > kMaxDynLds) return false;
  if (out) *out = s;
  return true;
}
// tallest tile the ring kernel's LDS layout holds
inline int conv_ring_max_rows(int cin, int cout, int K) {
  RingShape s;
  if (!conv_ring_shape(cin, cout, K, &s)) return 0;
  int T = 512;
  while (T >= 32 && conv_ring_lds(T, s, cin / 32, K, nullptr) > kMaxDynLds) T -= 16;
  return T >= 32 ? T : 0;
}

int launch_conv_ring6h(const ConvArgsH &a, int dtype, hipStream_t st);  // conv_ring_applies(); dtype 1 bf16, 2 fp16



typedef float ring_v2f __attribute__((ext_vector_type(2)));
typedef int ring_v4i __attribute__((ext_vector_type(4)));
typedef unsigned ring_v4u __attribute__((ext_vector_type(4)));  // native vectors: the HIP uint4 / float2 structs do not load across address spaces
#define PCS_LDS(T) __attribute__((address_space(3))) T

// LDS-DMA, 16 / 4 bytes per lane: lane l's bytes land at lds_dst + l * {16, 4}. M0 carries the (wave-uniform) LDS address
// and is written in the same statement that reads it; hipcc neither counts nor waits for these loads (the loader wave
// counts its own vmcnt).
__device__ __forceinline__ void glds16(const void *gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void glds4(const void *gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// workgroup barrier without the fence of __syncthreads() (which would drain the loader's DMA queue and every compute wave's
// weight prefetch): the compiler may not move memory operations across it, the hardware orders nothing but arrival
__device__ __forceinline__ void ring_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// ---- fp32 ring kernel (conv_ring6f.hip): shapes it serves and its LDS layout ----------------------------------------------------
// cin % 32 == 0; output columns in whole 16-column tiles, 4 (64 columns, one tile per compute wave), 6 or 8 (96 / 128 columns,
// two tiles per compute wave) per column tile; one or two 32-channel steps per weight chunk (cin / 32 even: two).
int &conv_ringf_mode();  // conv_ring6f.hip: 0 never, 1 wherever it applies, -1 per-shape policy (pcs_conv_ring_enable / PCS_CONV_RINGF)
inline int conv_ringf_nctt(int cout) {
  if (cout % 128 == 0) return 8;
  if (cout % 96 == 0) return 6;
  if (cout % 64 == 0) return 4;
  return 0;
}
inline bool conv_ringf_policy(int cin, int cout, int K) {
  (void)cin; (void)cout; (void)K;   // where it beats conv_os5_kernel (profiles/round4_ring.md)
  return false;
}
inline bool conv_ringf_shape(int cin, int cout, int K, RingShape *out) {
  const int mode = conv_ringf_mode();
  if (mode == 0 || cin % 32 || cin < 32 || K > 32 || K < 1) return false;
  const int nctt = conv_ringf_nctt(cout);
  if (!nctt) return false;
  if (mode < 0 && !conv_ringf_policy(cin, cout, K)) return false;
  const int ns = cin / 32, kc = ns % 2 == 0 ? 2 : 1;
  if (ns / kc > 15) return false;
  if (out) { out->nctt = nctt; out->nc = nctt == 4 ? 1 : 2; out->kc = kc; }
  return true;
}
// [A ring: kRingDepth batches of 2 row blocks x kc steps x 2 KB | pair-index ring | offset lists | batch descriptors | accumulator tile]
inline size_t conv_ringf_lds(int T, const RingShape &s, int ns, int K, int *acc_off) {
  size_t off = (size_t)kRingDepth * kRingBatchRows * s.kc * 2048 + (size_t)kRingMeta * kRingBatchRows * 128;
  off += 4 * 36 * 4 + 16;
  off = (off + 7) & ~(size_t)7;
  off += (size_t)conv_ring_bt_cap(T, ns, s.kc, K) * 8;
  off = (off + 15) & ~(size_t)15;
  if (acc_off) *acc_off = (int)off;
  return off + (size_t)(T + 1) * (16 * s.nctt + 4) * 4;
}
inline bool conv_ringf_applies(int cin, int cout, int K, int T, RingShape *out) {
  RingShape s;
  if (!conv_ringf_shape(cin, cout, K, &s)) return false;
  if (T < 32 || T > 512 || T % 16 || conv_ringf_lds(T, s, cin / 32, K, nullptr) > kMaxDynLds) return false;
  if (out) *out = s;
  return true;
}
inline int conv_ringf_max_rows(int cin, int cout, int K) {
  RingShape s;
  if (!conv_ringf_shape(cin, cout, K, &s)) return 0;
  int T = 512;
  while (T >= 32 && conv_ringf_lds(T, s, cin / 32, K, nullptr) > kMaxDynLds) T -= 16;
  return T >= 32 ? T : 0;
}
int launch_conv_ring6f(const ConvArgs &a, hipStream_t st);  // conv_ringf_applies()

}  // namespace pcs
