// Device helpers shared by the ring kernels (conv_ring6h.hip: 16-bit MFMA; conv_ring6f.hip: fp32 MFMA): LDS-DMA, the fence-less
// workgroup barrier, and the launch-shape rules of the fp32 ring kernel (the half kernel's live in conv_half.h).
#pragma once
#include "conv_half.h"

namespace pcs {

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
