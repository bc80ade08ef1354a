// Range image -> points: bilinear sampling of a (B, C, H, W) feature map at per-point image coordinates, forward and backward,
// for gfx950. Replaces what R:pcseg/model/segmentor/fusion/rpvnet/rpvnet.py:31-51 (`resample_grid_stacked` / `range_to_point`)
// does with a python loop over frames around `torch.nn.functional.grid_sample(mode='bilinear')` (zero padding,
// align_corners=False) -- same arithmetic: ix = ((x + 1) W - 1) / 2, corners nw / ne / sw / se, out-of-image corners contribute 0.
// Why: the BACKWARD of torch's op parallelises over the points and walks the channels with four float atomics each into NCHW planes
// -- 59.7 ms of a 235 ms RPVNet mk34 cr1.75 step (profiles/round5_config5_step_budget.md), 25 % of the step in one torch kernel.
// Both directions are HBM-bound gather / scatter passes; algorithmic bytes: forward 4 C (N + touched pixels) + 12 N, backward
// 4 C (N + B H W) + 16 N... + the corner table.
//   forward : a workgroup owns 64 consecutive points x a chunk of CH channels. Wave w samples channels w, w + 4, ... with the
//             lanes over the points (the four corner offsets / weights of a point live in registers, every plane read is a
//             4-byte gather out of L2) into an LDS tile [point][channel]; the tile is written as 16-byte row pieces of the
//             (N, C) output. One launch for all frames (the batch index is a column of pxpy), no per-frame masks.
//   backward: atomic-free and deterministic. The 4 N (point, corner) entries are keyed by their pixel (pcs_range_sample_corners)
//             and sorted once per step and resolution by the caller (a CSR over the B H W pixels, like the denselize passes of
//             scatter.hip); a workgroup owns 64 consecutive pixels of one image row x CH channels, sums w * gout[point] over each
//             pixel's entries with 16-byte row loads into an LDS tile and writes it transposed -- every NCHW store a full 256-byte
//             line, every element written exactly once (no memset).
#include "pcs_common.h"

using namespace pcs;

namespace {

constexpr int RS_PT = 64;   // points (forward) / pixels (backward) per workgroup

struct Corners {
  int64_t off[4];   // offset of the corner inside one (H, W) plane, -1 = outside the image
  float w[4];
};

// grid_sampler_compute_source_index + the four bilinear weights of torch's grid_sampler_2d (GridSampler.cuh), fp32 like there
__device__ __forceinline__ Corners corners_of(float x, float y, int H, int W) {
  const float ix = ((x + 1.f) * (float)W - 1.f) / 2.f;
  const float iy = ((y + 1.f) * (float)H - 1.f) / 2.f;
  const float fx = floorf(ix), fy = floorf(iy);
  const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
  Corners c;
  c.w[0] = ((float)x1 - ix) * ((float)y1 - iy);   // nw
  c.w[1] = (ix - (float)x0) * ((float)y1 - iy);   // ne
  c.w[2] = ((float)x1 - ix) * (iy - (float)y0);   // sw
  c.w[3] = (ix - (float)x0) * (iy - (float)y0);   // se
  const bool vx0 = x0 >= 0 && x0 < W, vx1 = x1 >= 0 && x1 < W, vy0 = y0 >= 0 && y0 < H, vy1 = y1 >= 0 && y1 < H;
  c.off[0] = (vx0 && vy0) ? (int64_t)y0 * W + x0 : -1;
  c.off[1] = (vx1 && vy0) ? (int64_t)y0 * W + x1 : -1;
  c.off[2] = (vx0 && vy1) ? (int64_t)y1 * W + x0 : -1;
  c.off[3] = (vx1 && vy1) ? (int64_t)y1 * W + x1 : -1;
  return c;
}

template <int CH>
__global__ void __launch_bounds__(256) range_sample_fwd_kernel(const float *__restrict__ img, const float *__restrict__ pxpy,
                                                               int64_t n, int B, int C, int H, int W, float *__restrict__ out) {
  __shared__ __attribute__((aligned(16))) float tile[RS_PT][CH + 4];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int64_t p0 = (int64_t)blockIdx.x * RS_PT;
  const int c0 = blockIdx.y * CH;
  const int64_t p = p0 + lane;
  Corners cn;
  int b = -1;
  if (p < n) {
    const float fb = pxpy[3 * p];
    b = (int)fb;
    if (!(fb >= 0.f) || b >= B || (float)b != fb) b = -1;   // a point of no frame samples nothing
    cn = corners_of(pxpy[3 * p + 1], pxpy[3 * p + 2], H, W);
  }
  const int64_t plane = (int64_t)H * W;
  for (int j = wid; j < CH; j += 4) {
    float v = 0.f;
    if (b >= 0 && c0 + j < C) {
      const float *pl = img + ((int64_t)b * C + c0 + j) * plane;
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (cn.off[k] >= 0) v += pl[cn.off[k]] * cn.w[k];   // nw, ne, sw, se: torch's order of accumulation
    }
    tile[lane][j] = v;
  }
  __syncthreads();
  constexpr int VL = CH / 4;
  for (int q = threadIdx.x; q < RS_PT * VL; q += 256) {
    const int r = q / VL, vl = q % VL;
    const int ch = c0 + 4 * vl;
    if (p0 + r < n && ch < C) *reinterpret_cast<float4 *>(out + (p0 + r) * C + ch) = *reinterpret_cast<const float4 *>(&tile[r][4 * vl]);
  }
}

// keys[4 p + k] = pixel of corner k of point p ((b H + y) W + x), -1 when the corner (or the point's frame) is outside; wts alongside
__global__ void __launch_bounds__(256) range_corners_kernel(const float *__restrict__ pxpy, int64_t n, int B, int H, int W,
                                                            int64_t *__restrict__ keys, float *__restrict__ wts) {
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x) {
    const float fb = pxpy[3 * p];
    int b = (int)fb;
    if (!(fb >= 0.f) || b >= B || (float)b != fb) b = -1;
    const Corners cn = corners_of(pxpy[3 * p + 1], pxpy[3 * p + 2], H, W);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      keys[4 * p + k] = (b >= 0 && cn.off[k] >= 0) ? (int64_t)b * H * W + cn.off[k] : -1;
      wts[4 * p + k] = cn.w[k];
    }
  }
}

template <int CH>
__global__ void __launch_bounds__(256) range_sample_bwd_kernel(const float *__restrict__ gout, const int64_t *__restrict__ order,
                                                               const int64_t *__restrict__ rowptr, const float *__restrict__ wts,
                                                               int B, int C, int H, int W, float *__restrict__ gimg) {
  constexpr int VL = CH / 4;     // lanes per pixel
  constexpr int PP = 256 / VL;   // pixels in flight
  __shared__ __attribute__((aligned(16))) float tile[RS_PT][CH + 4];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int xt = (W + RS_PT - 1) / RS_PT;
  const int x0 = (blockIdx.x % xt) * RS_PT;
  const int64_t row = blockIdx.x / xt;   // b * H + y
  const int c0 = blockIdx.y * CH;
  const int b = (int)(row / H), y = (int)(row % H);
  const int vl = threadIdx.x % VL, q0 = threadIdx.x / VL;
  const int ch = c0 + 4 * vl;
  for (int q = q0; q < RS_PT; q += PP) {
    const int x = x0 + q;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (x < W && ch < C) {
      const int64_t pos = row * W + x;
      const int64_t e1 = rowptr[pos + 1];
      for (int64_t e = rowptr[pos]; e < e1; ++e) {   // entries in ascending (point, corner) order: a fixed summation order
        const int64_t id = order[e];
        const float w = wts[id];
        const float4 g = *reinterpret_cast<const float4 *>(gout + (id >> 2) * C + ch);
        acc.x += g.x * w; acc.y += g.y * w; acc.z += g.z * w; acc.w += g.w * w;
      }
    }
    *reinterpret_cast<float4 *>(&tile[q][4 * vl]) = acc;
  }
  __syncthreads();
  if (x0 + lane < W)
    for (int j = wid; j < CH && c0 + j < C; j += 4)   // wave per channel, lanes over pixels
      gimg[(((int64_t)b * C + c0 + j) * H + y) * W + x0 + lane] = tile[lane][j];
}

}  // namespace

extern "C" int pcs_range_sample_fwd_f32(const float *img, const float *pxpy, int64_t n, int32_t B, int32_t C, int32_t H,
                                        int32_t W, float *out, void *stream) {
  if (n < 0 || B <= 0 || C <= 0 || H <= 0 || W <= 0) { set_error("pcs_range_sample_fwd: bad sizes"); return PCS_EINVAL; }
  if (n == 0) return PCS_OK;
  if (!img || !pxpy || !out) { set_error("pcs_range_sample_fwd: null pointer"); return PCS_EINVAL; }
  if ((C & 3) || ((uintptr_t)out & 15)) { set_error("pcs_range_sample_fwd: needs C % 4 == 0 and 16-byte aligned rows"); return PCS_EUNSUPPORTED; }
  const int64_t blocks = ceil_div(n, RS_PT);
  if (blocks > 0x7FFFFFFF) { set_error("pcs_range_sample_fwd: grid too large"); return PCS_EUNSUPPORTED; }
  hipStream_t st = as_stream(stream);
  if (C <= 32) hipLaunchKernelGGL(range_sample_fwd_kernel<32>, dim3((unsigned)blocks, 1), dim3(256), 0, st, img, pxpy, n, B, C, H, W, out);
  else hipLaunchKernelGGL(range_sample_fwd_kernel<64>, dim3((unsigned)blocks, (unsigned)ceil_div(C, 64)), dim3(256), 0, st, img, pxpy, n, B, C, H, W, out);
  return check_launch("pcs_range_sample_fwd");
}

extern "C" int pcs_range_sample_corners(const float *pxpy, int64_t n, int32_t B, int32_t H, int32_t W, int64_t *keys,
                                        float *wts, void *stream) {
  if (n < 0 || B <= 0 || H <= 0 || W <= 0) { set_error("pcs_range_sample_corners: bad sizes"); return PCS_EINVAL; }
  if (n == 0) return PCS_OK;
  if (!pxpy || !keys || !wts) { set_error("pcs_range_sample_corners: null pointer"); return PCS_EINVAL; }
  hipLaunchKernelGGL(range_corners_kernel, dim3(stream_grid(n, 256)), dim3(256), 0, as_stream(stream), pxpy, n, B, H, W, keys, wts);
  return check_launch("pcs_range_sample_corners");
}

extern "C" int pcs_range_sample_bwd_csr_f32(const float *gout, const int64_t *order, const int64_t *rowptr, const float *wts,
                                            int32_t B, int32_t C, int32_t H, int32_t W, float *gimg, void *stream) {
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || !rowptr || !gimg) { set_error("pcs_range_sample_bwd_csr: bad args"); return PCS_EINVAL; }
  if ((C & 3) || ((uintptr_t)gout & 15)) { set_error("pcs_range_sample_bwd_csr: needs C % 4 == 0 and 16-byte aligned rows"); return PCS_EUNSUPPORTED; }
  const int64_t blocks = (int64_t)B * H * ceil_div(W, RS_PT);
  if (blocks > 0x7FFFFFFF) { set_error("pcs_range_sample_bwd_csr: grid too large"); return PCS_EUNSUPPORTED; }
  hipStream_t st = as_stream(stream);
  if (C <= 32) hipLaunchKernelGGL(range_sample_bwd_kernel<32>, dim3((unsigned)blocks, 1), dim3(256), 0, st, gout, order, rowptr, wts, B, C, H, W, gimg);
  else hipLaunchKernelGGL(range_sample_bwd_kernel<64>, dim3((unsigned)blocks, (unsigned)ceil_div(C, 64)), dim3(256), 0, st, gout, order, rowptr, wts, B, C, H, W, gimg);
  return check_launch("pcs_range_sample_bwd_csr");
}
