// Fused training-mode BatchNorm (+ residual add + ReLU) over (N, C) voxel features -- gfx950, fp32, HBM-bound.
// The reference runs nn.BatchNorm1d / SyncBatchNorm, a separate ReLU and a separate residual add through
// fapply (R:pcseg/model/segmentor/voxel/minkunet/minkunet.py:23-129): 4 BN kernels + 2 ReLU + 2 add kernels per
// block and direction. Here: forward = one statistics pass + one apply pass (normalise, +residual, ReLU,
// single write); backward = one reduction pass + one apply pass producing dx and (optionally) the residual grad.
// Statistics are two-level (per-workgroup partials, then a fixed-order reduction): deterministic, and the
// (sum, sumsq) vector is what a multi-GPU run all-reduces between the two kernels (SyncBN semantics).
#include <type_traits>

#include "pcs_common.h"

using namespace pcs;

namespace {

constexpr int kStatBlocks = 1024;  // partial rows; each workgroup strides over the feature rows

// storage format of the feature tensors (x, residual, y, dy, dx, dres): fp32, or bf16 / fp16 under mixed precision
// (statistics, scale / shift and all arithmetic stay fp32 / double)
struct F32 {};
struct B16 {};
struct H16 {};
__device__ __forceinline__ float h2f(B16, uint16_t h) { return __uint_as_float((uint32_t)h << 16); }
__device__ __forceinline__ float h2f(H16, uint16_t h) { return (float)__builtin_bit_cast(_Float16, h); }
__device__ __forceinline__ uint16_t f2h(B16, float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (uint16_t)((u >> 16) | 0x40u);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
__device__ __forceinline__ uint16_t f2h(H16, float f) { const _Float16 h = (_Float16)f; return __builtin_bit_cast(uint16_t, h); }
// element `e` (a multiple of V) of a tensor -> V floats; and back
template <int V> __device__ __forceinline__ typename std::conditional<V == 4, float4, float>::type ldv(F32, const void *p, int64_t e) {
  return *reinterpret_cast<const typename std::conditional<V == 4, float4, float>::type *>(reinterpret_cast<const float *>(p) + e);
}
template <int V, typename HT> __device__ __forceinline__ typename std::conditional<V == 4, float4, float>::type ldv(HT, const void *p, int64_t e) {
  const uint16_t *h = reinterpret_cast<const uint16_t *>(p) + e;
  if constexpr (V == 4) {
    const uint2 r = *reinterpret_cast<const uint2 *>(h);
    return make_float4(h2f(HT{}, (uint16_t)(r.x & 0xFFFFu)), h2f(HT{}, (uint16_t)(r.x >> 16)),
                       h2f(HT{}, (uint16_t)(r.y & 0xFFFFu)), h2f(HT{}, (uint16_t)(r.y >> 16)));
  } else {
    return h2f(HT{}, h[0]);
  }
}
__device__ __forceinline__ void stv(F32, void *p, int64_t e, const float4 &v) { *reinterpret_cast<float4 *>(reinterpret_cast<float *>(p) + e) = v; }
__device__ __forceinline__ void stv(F32, void *p, int64_t e, const float &v) { reinterpret_cast<float *>(p)[e] = v; }
template <typename HT> __device__ __forceinline__ void stv(HT, void *p, int64_t e, const float4 &v) {
  uint2 o;
  o.x = f2h(HT{}, v.x) | ((uint32_t)f2h(HT{}, v.y) << 16);
  o.y = f2h(HT{}, v.z) | ((uint32_t)f2h(HT{}, v.w) << 16);
  *reinterpret_cast<uint2 *>(reinterpret_cast<uint16_t *>(p) + e) = o;
}
template <typename HT> __device__ __forceinline__ void stv(HT, void *p, int64_t e, const float &v) {
  reinterpret_cast<uint16_t *>(p)[e] = f2h(HT{}, v);
}

template <int V> struct NV;
template <> struct NV<4> { using T = float4; };
template <> struct NV<1> { using T = float; };
__device__ __forceinline__ float comp(const float4 &v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : (i == 2 ? v.z : v.w)); }
__device__ __forceinline__ float comp(const float &v, int) { return v; }
__device__ __forceinline__ void setc(float4 &v, int i, float s) { if (i == 0) v.x = s; else if (i == 1) v.y = s; else if (i == 2) v.z = s; else v.w = s; }
__device__ __forceinline__ void setc(float &v, int, float s) { v = s; }


// block = (TX lanes over channel VECTORS, TY rows); partial[b][0][c] = sum x, partial[b][1][c] = sum x^2
// (backward: sum g, sum g*xhat with g = dy * [y > 0]). V = 4: 16-byte loads.
template <bool BWD, int V, typename ET>
__global__ void __launch_bounds__(256) bn_partial_kernel(const void *__restrict__ x, const void *__restrict__ dy,
                                                         const void *__restrict__ y, const uint32_t *__restrict__ mask,
                                                         const double *__restrict__ stat,
                                                         int64_t n, int c, int cv, int relu, float *__restrict__ partial,
                                                         int64_t lddy) {
  using VT = typename NV<V>::T;
  extern __shared__ float red[];  // [TY][2][TX*V]
  const int tx = threadIdx.x, ty = threadIdx.y, TX = blockDim.x, TY = blockDim.y;
  const int W = TX * V;
  for (int j0 = 0; j0 < cv; j0 += TX) {
    const int j = j0 + tx;
    float s0[V], s1[V];
#pragma unroll
    for (int q = 0; q < V; ++q) { s0[q] = 0.f; s1[q] = 0.f; }
    if (j < cv) {
      // forward: sums of (x - pivot) and (x - pivot)^2 with pivot = row 0 of the tensor (a sample of the channel),
      // un-shifted in double by the reduce kernel: E[x^2] - mean^2 on RAW fp32 partial sums loses var/mean^2 digits
      float mean[V], invstd[V];
#pragma unroll
      for (int q = 0; q < V; ++q) {
        mean[q] = BWD ? (float)stat[j * V + q] : (n > 0 ? ldv<1>(ET{}, x, j * V + q) : 0.f);
        invstd[q] = BWD ? (float)stat[c + j * V + q] : 0.f;
        // forward: workgroup 0 leaves the pivot row, widened to fp32, in partial row gridDim.x for the reduce kernel
        if (!BWD && blockIdx.x == 0 && ty == 0) partial[(int64_t)gridDim.x * 2 * c + j * V + q] = mean[q];
      }
#pragma unroll 4  // four rows' loads in flight per thread: one 8/16-byte load per array and iteration left the pass latency-bound
      for (int64_t i = (int64_t)blockIdx.x * TY + ty; i < n; i += (int64_t)gridDim.x * TY) {
        const VT xv = ldv<V>(ET{}, x, i * c + (int64_t)j * V);
        if (BWD) {
          const VT gv = ldv<V>(ET{}, dy, i * lddy + (int64_t)j * V);
          VT yv; unsigned bits = 0xFu;
          if (relu) {
            if (V == 4 && mask) bits = (mask[i * (c >> 5) + (j >> 3)] >> (4 * (j & 7))) & 0xFu;
            else yv = ldv<V>(ET{}, y, i * c + (int64_t)j * V);
          }
#pragma unroll
          for (int q = 0; q < V; ++q) {
            float g = comp(gv, q);
            if (relu && ((V == 4 && mask) ? !((bits >> q) & 1u) : comp(yv, q) <= 0.f)) g = 0.f;
            s0[q] += g;
            s1[q] += g * ((comp(xv, q) - mean[q]) * invstd[q]);
          }
        } else {
#pragma unroll
          for (int q = 0; q < V; ++q) { const float t = comp(xv, q) - mean[q]; s0[q] += t; s1[q] += t * t; }
        }
      }
    }
#pragma unroll
    for (int q = 0; q < V; ++q) {
      red[(ty * 2 + 0) * W + tx * V + q] = s0[q];
      red[(ty * 2 + 1) * W + tx * V + q] = s1[q];
    }
    __syncthreads();
    if (ty == 0 && j < cv) {
#pragma unroll
      for (int q = 0; q < V; ++q) {
        float a = 0.f, b = 0.f;
        for (int r = 0; r < TY; ++r) { a += red[(r * 2 + 0) * W + tx * V + q]; b += red[(r * 2 + 1) * W + tx * V + q]; }
        partial[((int64_t)blockIdx.x * 2 + 0) * c + j * V + q] = a;
        partial[((int64_t)blockIdx.x * 2 + 1) * c + j * V + q] = b;
      }
    }
    __syncthreads();
  }
}

// sums[ch] / sums[c + ch] = sum over the nblk partial rows of partial[b][0][ch] / partial[b][1][ch], accumulated in
// double in a FIXED order (lane ty sums rows ty, ty+256, ...; groups of 16 lanes in order; the 16 group sums in order):
// deterministic. The kernel is pure latency (a few MB once per BatchNorm pass, ~250 launches per training step), so the
// rows are spread over as many lanes as a workgroup holds and each lane keeps its loads in flight together. Forward (pivot != NULL): the partials are sums of (x - pivot) and (x - pivot)^2; the raw moments
//   sum x = S0 + n p,   sum x^2 = S1 + 2 p S0 + n p^2
// are formed here in double, and sums[2c] = n (the vector a data-parallel run all-reduces: SyncBN needs the global count).
// Optional tails of the same launch (each channel's totals sit in one thread, so neither needs another kernel):
//   f32copy (2c floats): the sums again in fp32 -- backward: the weight / bias gradients in the parameters' dtype;
//   stat (2c doubles):   bn_finalize_kernel's mean / invstd + running statistics with count = n -- the forward of a
//                        single-process run, where nothing has to be all-reduced between the reduction and the finalize.
constexpr int kRedCh = 4;       // channels per workgroup of bn_reduce_kernel
constexpr int kRedLanes = 256;  // row lanes per workgroup
template <typename PT>
__global__ void __launch_bounds__(1024) bn_reduce_kernel(const PT *__restrict__ partial, int nblk, int c,
                                                         const float *__restrict__ pivot, int64_t n,
                                                         double *__restrict__ sums, int write_count,
                                                         float *__restrict__ f32copy = nullptr, double *__restrict__ stat = nullptr,
                                                         double eps = 0.0, double momentum = 0.0, float *running_mean = nullptr,
                                                         float *running_var = nullptr) {
  // 4 channels x 256 row lanes per workgroup (round 4; was 16 x 64): c / 4 workgroups instead of c / 16 -- a 96-channel
  // layer's 3 000 conv-tile partial rows went through 6 CUs (32 us), the 1 024 rows of the BatchNorm passes took 15 us
  // whatever c: the kernel is latency, so it is spread over more CUs and every lane issues all its loads at once.
  __shared__ double red[2][kRedLanes][kRedCh + 1];
  __shared__ double red2[2][16][kRedCh + 1];
  const int ch = blockIdx.x * kRedCh + threadIdx.x;
  double s0 = 0.0, s1 = 0.0;
  if (ch < c) {
    PT v0[8], v1[8];
    for (int b0 = threadIdx.y; b0 < nblk; b0 += kRedLanes * 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int b = b0 + kRedLanes * u;
        v0[u] = b < nblk ? partial[(int64_t)b * 2 * c + ch] : (PT)0;
        v1[u] = b < nblk ? partial[(int64_t)b * 2 * c + c + ch] : (PT)0;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) { s0 += (double)v0[u]; s1 += (double)v1[u]; }
    }
  }
  red[0][threadIdx.y][threadIdx.x] = s0;
  red[1][threadIdx.y][threadIdx.x] = s1;
  __syncthreads();
  if (threadIdx.y < 16) {   // fixed order: 16 groups of 16 lanes, then the 16 group sums
    double a = 0.0, b = 0.0;
    for (int r = 0; r < 16; ++r) { a += red[0][threadIdx.y * 16 + r][threadIdx.x]; b += red[1][threadIdx.y * 16 + r][threadIdx.x]; }
    red2[0][threadIdx.y][threadIdx.x] = a;
    red2[1][threadIdx.y][threadIdx.x] = b;
  }
  __syncthreads();
  if (threadIdx.y == 0 && ch < c) {
    double t0 = 0.0, t1 = 0.0;
    for (int r = 0; r < 16; ++r) { t0 += red2[0][r][threadIdx.x]; t1 += red2[1][r][threadIdx.x]; }
    if (pivot) {
      const double p = n > 0 ? (double)pivot[ch] : 0.0, dn = (double)n;
      t1 = t1 + 2.0 * p * t0 + dn * p * p;
      t0 = t0 + dn * p;
    }
    if (sums) {
      sums[ch] = t0;
      sums[c + ch] = t1;
    }
    if (f32copy) {
      f32copy[ch] = (float)t0;
      f32copy[c + ch] = (float)t1;
    }
    if (stat) {  // bn_finalize_kernel below, count = n
      const double count = n > 0 ? (double)n : 1.0;
      const double mean = t0 / count;
      double var = t1 / count - mean * mean;
      if (var < 0.0) var = 0.0;
      stat[ch] = mean;
      stat[c + ch] = 1.0 / sqrt(var + eps);
      if (running_mean) {
        const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
        running_mean[ch] = (float)((1.0 - momentum) * running_mean[ch] + momentum * mean);
        running_var[ch] = (float)((1.0 - momentum) * running_var[ch] + momentum * unb);
      }
    }
  }
  if (sums && write_count && blockIdx.x == 0 && threadIdx.x == 0 && threadIdx.y == 0) sums[2 * c] = (double)n;
}

// stat[0..c) = mean, stat[c..2c) = invstd; running stats updated like nn.BatchNorm1d (unbiased var, momentum)
__global__ void __launch_bounds__(256) bn_finalize_kernel(const double *__restrict__ sums, double count,
                                                          const double *__restrict__ count_dev, int c,
                                                          double eps, double momentum, float *running_mean,
                                                          float *running_var, double *__restrict__ stat) {
  const int ch = blockIdx.x * blockDim.x + threadIdx.x;
  if (ch >= c) return;
  if (count_dev) count = *count_dev;  // the all-reduced global row count stays on the device (no host read)
  if (!(count > 0.0)) count = 1.0;
  const double mean = sums[ch] / count;
  double var = sums[c + ch] / count - mean * mean;
  if (var < 0.0) var = 0.0;
  stat[ch] = mean;
  stat[c + ch] = 1.0 / sqrt(var + eps);
  if (running_mean) {
    const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
    running_mean[ch] = (float)((1.0 - momentum) * running_mean[ch] + momentum * mean);
    running_var[ch] = (float)((1.0 - momentum) * running_var[ch] + momentum * unb);
  }
}

// y = act((x - mean) * invstd * w + b [+ res])
template <int V, typename ET>
__global__ void __launch_bounds__(256) bn_apply_kernel(const void *__restrict__ x, const void *__restrict__ res,
                                                       const double *__restrict__ stat, const float *__restrict__ w,
                                                       const float *__restrict__ b, int64_t n, int c, int cv, int relu,
                                                       void *__restrict__ y, uint32_t *__restrict__ mask, int64_t ldy,
                                                       const void *__restrict__ tail, int ctail) {
  using VT = typename NV<V>::T;
  // concat fusion (torchsparse.cat([bn_relu(conv(x)), skip])): y is the left c columns of a (n, ldy) buffer and the
  // skip tensor `tail` (n, ctail) is copied into the columns right of it by the same launch
  for (int j = threadIdx.x; j < ctail / V; j += blockDim.x) {
#pragma unroll 4
    for (int64_t i = (int64_t)blockIdx.x * blockDim.y + threadIdx.y; i < n; i += (int64_t)gridDim.x * blockDim.y)
      stv(ET{}, y, i * ldy + c + (int64_t)j * V, ldv<V>(ET{}, tail, i * ctail + (int64_t)j * V));
  }
  for (int j = threadIdx.x; j < cv; j += blockDim.x) {
    float sc[V], sh[V];
#pragma unroll
    for (int q = 0; q < V; ++q) {
      const int ch = j * V + q;
      const float invstd = (float)stat[c + ch], mean = (float)stat[ch];
      sc[q] = invstd * (w ? w[ch] : 1.f);
      sh[q] = (b ? b[ch] : 0.f) - mean * sc[q];
    }
#pragma unroll 4
    for (int64_t i = (int64_t)blockIdx.x * blockDim.y + threadIdx.y; i < n; i += (int64_t)gridDim.x * blockDim.y) {
      const VT xv = ldv<V>(ET{}, x, i * c + (int64_t)j * V);
      VT rv; if (res) rv = ldv<V>(ET{}, res, i * c + (int64_t)j * V);
      VT o;
      unsigned bits = 0;
#pragma unroll
      for (int q = 0; q < V; ++q) {
        float t = fmaf(comp(xv, q), sc[q], sh[q]);
        if (res) t += comp(rv, q);
        if (relu && t < 0.f) t = 0.f;
        bits |= (t > 0.f ? 1u : 0u) << q;
        setc(o, q, t);
      }
      stv(ET{}, y, i * ldy + (int64_t)j * V, o);
      if (V == 4 && mask) {  // c % 32 == 0: 8 consecutive lanes (4 channels each) of one row make one word
        unsigned m = bits << (4 * (j & 7));
        m |= __shfl_xor(m, 1, 64); m |= __shfl_xor(m, 2, 64); m |= __shfl_xor(m, 4, 64);
        if ((j & 7) == 0) mask[i * (c >> 5) + (j >> 3)] = m;
      }
    }
  }
}

// g = dy * [y > 0];  dx = (g - sum_g/N - xhat * sum_gxhat/N) * invstd * w ;  dres = g
template <int V, typename ET>
__global__ void __launch_bounds__(256) bn_bwd_apply_kernel(const void *__restrict__ dy, const void *__restrict__ x,
                                                           const void *__restrict__ y, const uint32_t *__restrict__ mask,
                                                           const double *__restrict__ stat,
                                                           const double *__restrict__ sums2, double count,
                                                           const double *__restrict__ count_dev,
                                                           const float *__restrict__ w, int64_t n, int c, int cv,
                                                           int relu, void *__restrict__ dx, void *__restrict__ dres,
                                                           int64_t lddy, float in_slope) {
  // in_slope != 1: the BatchNorm's INPUT x is a LeakyReLU output (conv -> LeakyReLU -> BatchNorm,
  // R:pcseg/model/segmentor/voxel/cylinder3d/cylinder_ts.py:88-190): dx is multiplied by the activation's derivative
  // (x > 0 ? 1 : in_slope; the sign of a LeakyReLU output is its input's), i.e. it leaves as the gradient of the PRE-activation
  using VT = typename NV<V>::T;
  if (count_dev) count = *count_dev;
  if (!(count > 0.0)) count = 1.0;
  for (int j = threadIdx.x; j < cv; j += blockDim.x) {
    float mean[V], invstd[V], k1[V], k2[V], ws[V];
#pragma unroll
    for (int q = 0; q < V; ++q) {
      const int ch = j * V + q;
      mean[q] = (float)stat[ch]; invstd[q] = (float)stat[c + ch];
      k1[q] = (float)(sums2[ch] / count);
      k2[q] = (float)(sums2[c + ch] / count);
      ws[q] = invstd[q] * (w ? w[ch] : 1.f);
    }
#pragma unroll 4
    for (int64_t i = (int64_t)blockIdx.x * blockDim.y + threadIdx.y; i < n; i += (int64_t)gridDim.x * blockDim.y) {
      const VT gv = ldv<V>(ET{}, dy, i * lddy + (int64_t)j * V);
      const VT xv = ldv<V>(ET{}, x, i * c + (int64_t)j * V);
      VT yv; unsigned bits = 0xFu;
      if (relu) {
        if (V == 4 && mask) bits = (mask[i * (c >> 5) + (j >> 3)] >> (4 * (j & 7))) & 0xFu;
        else yv = ldv<V>(ET{}, y, i * c + (int64_t)j * V);
      }
      VT o, r;
#pragma unroll
      for (int q = 0; q < V; ++q) {
        float g = comp(gv, q);
        if (relu && ((V == 4 && mask) ? !((bits >> q) & 1u) : comp(yv, q) <= 0.f)) g = 0.f;
        const float xh = (comp(xv, q) - mean[q]) * invstd[q];
        float d = (g - k1[q] - xh * k2[q]) * ws[q];
        if (in_slope != 1.f && !(comp(xv, q) > 0.f)) d *= in_slope;
        setc(o, q, d);
        setc(r, q, g);
      }
      stv(ET{}, dx, i * c + (int64_t)j * V, o);
      if (dres) stv(ET{}, dres, i * c + (int64_t)j * V, r);
    }
  }
}

struct Geo { dim3 block, grid; int cv; };
template <int V> Geo geo(int64_t n, int c) {
  Geo g; g.cv = c / V;
  // one x-lane per channel vector up to 64 (NOT rounded to a power of two: 96 channels = 24 vectors would idle 8 of 32
  // lanes on the two widest levels); rows of a workgroup are contiguous in memory either way. The ReLU-mask shuffles of
  // bn_apply_kernel work on aligned groups of 8 lanes: c % 32 == 0 there, so cv % 8 == 0 and the groups stay aligned.
  const int tx = g.cv < 64 ? (g.cv > 0 ? g.cv : 1) : 64;
  g.block = dim3(tx, 256 / tx);
  int64_t gr = ceil_div(n > 0 ? n : 1, (256 / tx) * 4);
  if (gr > 2048) gr = 2048;
  g.grid = dim3((unsigned)gr);
  return g;
}

}  // namespace

extern "C" int32_t pcs_bn_num_partials(void) { return kStatBlocks + 1; }  // + the widened pivot row

// dtype of the feature tensors: 0 fp32, 1 bf16, 2 fp16. alignment unit of a V = 4 access: 16 B (fp32) / 8 B (halfs)
#define PCS_BN_DISPATCH(DT, CALL)                         \
  do {                                                    \
    if ((DT) == 0) { using ET = F32; CALL; }              \
    else if ((DT) == 1) { using ET = B16; CALL; }         \
    else { using ET = H16; CALL; }                        \
  } while (0)

static bool al_v4(int dtype, const void *p) { return ((uintptr_t)p & (dtype == 0 ? 15 : 7)) == 0; }

static int bn_partial(bool bwd, int dtype, const void *x, const void *dy, const void *y, const uint32_t *mask, const double *stat,
                      int64_t n, int c, int relu, float *partial, double *sums, hipStream_t st, int64_t lddy = 0) {
  if (lddy == 0) lddy = c;
  if (lddy < c) { set_error("pcs_bn: row stride of dy smaller than c"); return PCS_EINVAL; }
  const bool vec = (c & 3) == 0 && (lddy & 3) == 0 && al_v4(dtype, x) && al_v4(dtype, dy) && al_v4(dtype, y);
  if (mask && (!vec || (c & 31))) { set_error("pcs_bn: the ReLU bit mask needs c % 32 == 0 and aligned rows"); return PCS_EUNSUPPORTED; }
  const int V = vec ? 4 : 1, cv = c / V;
  const int tx = cv < 64 ? cv : 64;   // as geo(): one x-lane per channel vector, no power-of-two rounding
  dim3 block(tx, 256 / tx);
  const size_t lds = (size_t)(256 / tx) * 2 * tx * V * sizeof(float);
  if (vec) {
    if (bwd) PCS_BN_DISPATCH(dtype, hipLaunchKernelGGL((bn_partial_kernel<true, 4, ET>), dim3(kStatBlocks), block, lds, st, x, dy, y, mask, stat, n, c, cv, relu, partial, lddy));
    else PCS_BN_DISPATCH(dtype, hipLaunchKernelGGL((bn_partial_kernel<false, 4, ET>), dim3(kStatBlocks), block, lds, st, x, dy, y, mask, stat, n, c, cv, relu, partial, lddy));
  } else {
    if (bwd) PCS_BN_DISPATCH(dtype, hipLaunchKernelGGL((bn_partial_kernel<true, 1, ET>), dim3(kStatBlocks), block, lds, st, x, dy, y, mask, stat, n, c, cv, relu, partial, lddy));
    else PCS_BN_DISPATCH(dtype, hipLaunchKernelGGL((bn_partial_kernel<false, 1, ET>), dim3(kStatBlocks), block, lds, st, x, dy, y, mask, stat, n, c, cv, relu, partial, lddy));
  }
    const float *pivot = bwd ? nullptr : partial + (size_t)kStatBlocks * 2 * c;  // written by workgroup 0 above
  // backward: the sums once more in fp32 behind the 2c doubles (the parameter gradients, no conversion launch)
  hipLaunchKernelGGL(bn_reduce_kernel<float>, dim3((unsigned)ceil_div(c, kRedCh)), dim3(kRedCh, kRedLanes), 0, st, partial, kStatBlocks, c,
                     pivot, n, sums, bwd ? 0 : 1, bwd ? reinterpret_cast<float *>(sums + 2 * (size_t)c) : (float *)nullptr,
                     (double *)nullptr, 0.0, 0.0, (float *)nullptr, (float *)nullptr);
  return check_launch("pcs_bn_partial");
}

static int bn_apply_any(int dtype, const void *x, const void *res, const double *stat, const float *w, const float *b,
                        int64_t n, int32_t c, int32_t relu, void *y, uint32_t *mask, int64_t ldy, const void *tail,
                        int32_t ctail, void *stream) {
  if (n < 0 || c <= 0 || ctail < 0) { set_error("pcs_bn_apply: bad sizes"); return PCS_EINVAL; }
  if (n == 0) return PCS_OK;
  if (!x || !stat || !y || (ctail > 0 && !tail)) { set_error("pcs_bn_apply: null pointer"); return PCS_EINVAL; }
  if (ldy == 0) ldy = c;
  if (ldy < (int64_t)c + ctail) { set_error("pcs_bn_apply: row stride of y smaller than c + ctail"); return PCS_EINVAL; }
  hipStream_t st = as_stream(stream);
  const bool vec = (c & 3) == 0 && (ldy & 3) == 0 && (ctail & 3) == 0 && al_v4(dtype, x) && al_v4(dtype, y) && al_v4(dtype, res) &&
                   al_v4(dtype, tail);
  if (mask && (!vec || (c & 31))) { set_error("pcs_bn_apply: the ReLU bit mask needs c % 32 == 0 and aligned rows"); return PCS_EUNSUPPORTED; }
  if (vec) {
    Geo g = geo<4>(n, c);
    PCS_BN_DISPATCH(dtype, hipLaunchKernelGGL((bn_apply_kernel<4, ET>), g.grid, g.block, 0, st, x, res, stat, w, b, n, c, g.cv, relu, y, mask, ldy, tail, ctail));
  } else {
    Geo g = geo<1>(n, c);
    PCS_BN_DISPATCH(dtype, hipLaunchKernelGGL((bn_apply_kernel<1, ET>), g.grid, g.block, 0, st, x, res, stat, w, b, n, c, g.cv, relu, y, mask, ldy, tail, ctail));
  }
  return check_launch("pcs_bn_apply");
}

static int bn_bwd_apply_any(int dtype, const void *dy, const void *x, const void *y, const uint32_t *mask,
                            const double *stat, const double *sums2, double count, const double *count_dev,
                            const float *w, int64_t n, int32_t c, int32_t relu, void *dx, void *dres, int64_t lddy,
                            void *stream, float in_slope = 1.f) {
  if (n < 0 || c <= 0 || (!count_dev && !(count > 0))) { set_error("pcs_bn_bwd_apply: bad sizes"); return PCS_EINVAL; }
  if (lddy == 0) lddy = c;
  if (lddy < c) { set_error("pcs_bn_bwd_apply: row stride of dy smaller than c"); return PCS_EINVAL; }
  if (n == 0) return PCS_OK;
  if (!dy || !x || !stat || !sums2 || !dx || (relu && !y && !mask)) { set_error("pcs_bn_bwd_apply: null pointer"); return PCS_EINVAL; }
  hipStream_t st = as_stream(stream);
  const bool vec = (c & 3) == 0 && (lddy & 3) == 0 && al_v4(dtype, dy) && al_v4(dtype, x) && al_v4(dtype, y) && al_v4(dtype, dx) && al_v4(dtype, dres);
  if (mask && (!vec || (c & 31))) { set_error("pcs_bn_bwd_apply: the ReLU bit mask needs c % 32 == 0 and aligned rows"); return PCS_EUNSUPPORTED; }
  if (vec) {
    Geo g = geo<4>(n, c);
    PCS_BN_DISPATCH(dtype, hipLaunchKernelGGL((bn_bwd_apply_kernel<4, ET>), g.grid, g.block, 0, st, dy, x, y, mask, stat, sums2, count, count_dev, w, n, c, g.cv, relu, dx, dres, lddy, in_slope));
  } else {
    Geo g = geo<1>(n, c);
    PCS_BN_DISPATCH(dtype, hipLaunchKernelGGL((bn_bwd_apply_kernel<1, ET>), g.grid, g.block, 0, st, dy, x, y, mask, stat, sums2, count, count_dev, w, n, c, g.cv, relu, dx, dres, lddy, in_slope));
  }
  return check_launch("pcs_bn_bwd_apply");
}

static bool bad_half(int32_t dtype) { return dtype != 1 && dtype != 2; }

extern "C" int pcs_bn_stats_f32(const float *x, int64_t n, int32_t c, float *partial_ws, double *sums, void *stream) {
  if (n < 0 || c <= 0 || !x || !partial_ws || !sums) { set_error("pcs_bn_stats: bad args"); return PCS_EINVAL; }
  return bn_partial(false, 0, x, nullptr, nullptr, nullptr, nullptr, n, c, 0, partial_ws, sums, as_stream(stream));
}
extern "C" int pcs_bn_stats_h(const void *x, int64_t n, int32_t c, int32_t dtype, float *partial_ws, double *sums, void *stream) {
  if (n < 0 || c <= 0 || !x || !partial_ws || !sums || bad_half(dtype)) { set_error("pcs_bn_stats_h: bad args"); return PCS_EINVAL; }
  return bn_partial(false, dtype, x, nullptr, nullptr, nullptr, nullptr, n, c, 0, partial_ws, sums, as_stream(stream));
}

// sums (2c + 1) from the per-tile double partials a fused convolution left in its write-back
// (pcs_conv_gather_gemm_*'s bn_partial: [nrows][2][c] raw sums) -- replaces the pcs_bn_stats_* pass over the tensor
extern "C" int pcs_bn_reduce_partials(const double *partial, int64_t nrows, int32_t c, int64_t n, double *sums, void *stream) {
  if (nrows < 0 || nrows > 0x7FFFFFFF || c <= 0 || n < 0 || !partial || !sums) { set_error("pcs_bn_reduce_partials: bad args"); return PCS_EINVAL; }
  hipLaunchKernelGGL(bn_reduce_kernel<double>, dim3((unsigned)ceil_div(c, kRedCh)), dim3(kRedCh, kRedLanes), 0, as_stream(stream), partial,
                     (int)nrows, c, (const float *)nullptr, n, sums, 1, (float *)nullptr, (double *)nullptr, 0.0, 0.0,
                     (float *)nullptr, (float *)nullptr);
  return check_launch("pcs_bn_reduce_partials");
}

// backward: sums2 = [sum g | sum g xhat] (2c doubles, then the same 2c values as floats: the parameter gradients) from the per-tile
// partials a dgrad launch left in its write-back (pcs_conv_gather_gemm_*_ex with bn_x) -- replaces the pcs_bn_bwd_stats_* pass
extern "C" int pcs_bn_bwd_reduce_partials(const double *partial, int64_t nrows, int32_t c, double *sums2, int64_t sums2_doubles,
                                          void *stream) {
  if (nrows < 0 || nrows > 0x7FFFFFFF || c <= 0 || !partial || !sums2) { set_error("pcs_bn_bwd_reduce_partials: bad args"); return PCS_EINVAL; }
  if (sums2_doubles < 3 * (int64_t)c) { set_error("pcs_bn_bwd_reduce_partials: sums2 needs 3 c doubles"); return PCS_EWORKSPACE; }
  hipLaunchKernelGGL(bn_reduce_kernel<double>, dim3((unsigned)ceil_div(c, kRedCh)), dim3(kRedCh, kRedLanes), 0, as_stream(stream), partial,
                     (int)nrows, c, (const float *)nullptr, (int64_t)0, sums2, 0, reinterpret_cast<float *>(sums2 + 2 * (size_t)c),
                     (double *)nullptr, 0.0, 0.0, (float *)nullptr, (float *)nullptr);
  return check_launch("pcs_bn_bwd_reduce_partials");
}

// the same reduction with pcs_bn_finalize_f32 (count = n) in its tail: stat (2c) from the convolution's partials in ONE launch.
// sums (2c + 1) may be NULL. Not for SyncBN (the sums of all ranks must be added between the two steps).
extern "C" int pcs_bn_reduce_partials_finalize(const double *partial, int64_t nrows, int32_t c, int64_t n, double eps,
                                               double momentum, float *running_mean, float *running_var, double *sums,
                                               double *stat, void *stream) {
  if (nrows < 0 || nrows > 0x7FFFFFFF || c <= 0 || n <= 0 || !partial || !stat || (!running_mean) != (!running_var)) {
    set_error("pcs_bn_reduce_partials_finalize: bad args");
    return PCS_EINVAL;
  }
  hipLaunchKernelGGL(bn_reduce_kernel<double>, dim3((unsigned)ceil_div(c, kRedCh)), dim3(kRedCh, kRedLanes), 0, as_stream(stream), partial,
                     (int)nrows, c, (const float *)nullptr, n, sums, 1, (float *)nullptr, stat, eps, momentum, running_mean,
                     running_var);
  return check_launch("pcs_bn_reduce_partials_finalize");
}

extern "C" int pcs_bn_finalize_f32(const double *sums, double count, const double *count_dev, int32_t c, double eps,
                                   double momentum, float *running_mean, float *running_var, double *stat, void *stream) {
  if (c <= 0 || (!count_dev && !(count > 0)) || !sums || !stat) { set_error("pcs_bn_finalize: bad args"); return PCS_EINVAL; }
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((unsigned)ceil_div(c, 256)), dim3(256), 0, as_stream(stream), sums, count,
                     count_dev, c, eps, momentum, running_mean, running_var, stat);
  return check_launch("pcs_bn_finalize");
}

extern "C" int pcs_bn_apply_f32(const float *x, const float *res, const double *stat, const float *w, const float *b,
                                int64_t n, int32_t c, int32_t relu, float *y, uint32_t *mask, int64_t ldy, const float *tail,
                                int32_t ctail, void *stream) {
  return bn_apply_any(0, x, res, stat, w, b, n, c, relu, y, mask, ldy, tail, ctail, stream);
}
extern "C" int pcs_bn_apply_h(const void *x, const void *res, const double *stat, const float *w, const float *b,
                              int64_t n, int32_t c, int32_t relu, int32_t dtype, void *y, uint32_t *mask, int64_t ldy,
                              const void *tail, int32_t ctail, void *stream) {
  if (bad_half(dtype)) { set_error("pcs_bn_apply_h: dtype must be 1 (bf16) or 2 (fp16)"); return PCS_EINVAL; }
  return bn_apply_any(dtype, x, res, stat, w, b, n, c, relu, y, mask, ldy, tail, ctail, stream);
}

extern "C" int pcs_bn_bwd_stats_f32(const float *dy, const float *x, const float *y, const uint32_t *mask,
                                    const double *stat, int64_t n, int32_t c, int32_t relu, float *partial_ws,
                                    double *sums2, int64_t sums2_doubles, int64_t lddy, void *stream) {
  if (n < 0 || c <= 0 || !dy || !x || !stat || !partial_ws || !sums2 || (relu && !y && !mask)) { set_error("pcs_bn_bwd_stats: bad args"); return PCS_EINVAL; }
  if (sums2_doubles < 3 * (int64_t)c) { set_error("pcs_bn_bwd_stats: sums2 must hold 3c doubles (2c sums + the same 2c values as floats)"); return PCS_EWORKSPACE; }
  return bn_partial(true, 0, x, dy, y, mask, stat, n, c, relu, partial_ws, sums2, as_stream(stream), lddy);
}
extern "C" int pcs_bn_bwd_stats_h(const void *dy, const void *x, const void *y, const uint32_t *mask,
                                  const double *stat, int64_t n, int32_t c, int32_t relu, int32_t dtype, float *partial_ws,
                                  double *sums2, int64_t sums2_doubles, int64_t lddy, void *stream) {
  if (n < 0 || c <= 0 || !dy || !x || !stat || !partial_ws || !sums2 || (relu && !y && !mask) || bad_half(dtype)) { set_error("pcs_bn_bwd_stats_h: bad args"); return PCS_EINVAL; }
  if (sums2_doubles < 3 * (int64_t)c) { set_error("pcs_bn_bwd_stats_h: sums2 must hold 3c doubles (2c sums + the same 2c values as floats)"); return PCS_EWORKSPACE; }
  return bn_partial(true, dtype, x, dy, y, mask, stat, n, c, relu, partial_ws, sums2, as_stream(stream), lddy);
}

extern "C" int pcs_bn_bwd_apply_f32(const float *dy, const float *x, const float *y, const uint32_t *mask,
                                    const double *stat, const double *sums2, double count, const double *count_dev,
                                    const float *w, int64_t n, int32_t c, int32_t relu, float *dx, float *dres,
                                    int64_t lddy, void *stream) {
  return bn_bwd_apply_any(0, dy, x, y, mask, stat, sums2, count, count_dev, w, n, c, relu, dx, dres, lddy, stream);
}
// the same with the BatchNorm's input being a LeakyReLU output: dx leaves multiplied by (x > 0 ? 1 : in_slope) -- the gradient of the
// activation's INPUT, handed straight to the convolution that produced it (its write-back applied the LeakyReLU: pcs_conv_epilogue.act_slope)
extern "C" int pcs_bn_bwd_apply_act(const void *dy, const void *x, const void *y, const uint32_t *mask,
                                    const double *stat, const double *sums2, double count, const double *count_dev,
                                    const float *w, int64_t n, int32_t c, int32_t relu, int32_t dtype, float in_slope, void *dx,
                                    void *dres, int64_t lddy, void *stream) {
  if (dtype < 0 || dtype > 2 || !(in_slope > 0.f)) { set_error("pcs_bn_bwd_apply_act: dtype must be 0 / 1 / 2 and in_slope > 0"); return PCS_EINVAL; }
  return bn_bwd_apply_any(dtype, dy, x, y, mask, stat, sums2, count, count_dev, w, n, c, relu, dx, dres, lddy, stream, in_slope);
}
extern "C" int pcs_bn_bwd_apply_h(const void *dy, const void *x, const void *y, const uint32_t *mask,
                                  const double *stat, const double *sums2, double count, const double *count_dev,
                                  const float *w, int64_t n, int32_t c, int32_t relu, int32_t dtype, void *dx, void *dres,
                                  int64_t lddy, void *stream) {
  if (bad_half(dtype)) { set_error("pcs_bn_bwd_apply_h: dtype must be 1 (bf16) or 2 (fp16)"); return PCS_EINVAL; }
  return bn_bwd_apply_any(dtype, dy, x, y, mask, stat, sums2, count, count_dev, w, n, c, relu, dx, dres, lddy, stream);
}
