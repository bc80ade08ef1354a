// Every layer's weight preparation of a training step in ONE launch.
// The reference transposes / casts a layer's weights inside each backward (forward) call
// (TS:torchsparse/backend/convolution/convolution_cuda.cu:196-206: `kernel.transpose(1, 2)`; under --amp the half
// cast of `custom_fwd(cast_inputs=torch.half)`, TS:torchsparse/nn/functional/conv.py:19). Round 3 did the same per layer
// call: 55 pcs_transpose_kab_f32 launches per fp32 step, 113 pcs_conv_prepare_weights_h launches per bf16 step, each a few
// microseconds of work behind a launch. The weights only change at the optimizer step, so the host layer
// (functional._WeightPrep) refreshes ALL layers' prepared copies together the first time one is found stale: a job table on the
// device, one workgroup per 32 x 32 transpose tile / per 256 fragment words, workgroup -> job by binary search.
// Same element-wise results as the per-layer entry points (tests/test_hip_parity.py::test_weights_multi).
#include "conv_half.h"

using namespace pcs;

namespace {

// pcs_weight_job of include/pcseg_hip.h
struct Job {
  const float *src;
  void *dst;
  int32_t K, A, B, kind, transpose, nctt, nt16, ns;
  int64_t first_block;
};
static_assert(sizeof(Job) == sizeof(pcs_weight_job), "pcs_weight_job layout");

__device__ __forceinline__ void transpose_tile(const Job &j, int64_t lb, float (*tile)[33]) {
  const int tb = (j.B + 31) / 32, ta = (j.A + 31) / 32;
  const int bx = (int)(lb % tb);
  const int ay = (int)((lb / tb) % ta);
  const int k = (int)(lb / ((int64_t)tb * ta));
  const int a0 = ay * 32, b0 = bx * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const float *s = j.src + (int64_t)k * j.A * j.B;
  float *d = reinterpret_cast<float *>(j.dst) + (int64_t)k * j.A * j.B;
#pragma unroll
  for (int r = ty; r < 32; r += 8)
    if (a0 + r < j.A && b0 + tx < j.B) tile[r][tx] = s[(int64_t)(a0 + r) * j.B + b0 + tx];
  __syncthreads();
#pragma unroll
  for (int r = ty; r < 32; r += 8)
    if (b0 + r < j.B && a0 + tx < j.A) d[(int64_t)(b0 + r) * j.A + a0 + tx] = tile[tx][r];
}

// one fragment word (8 halfs) per thread: prepare_weights_kernel of conv_wave5h.hip, word i of the job
template <typename HT>
__device__ __forceinline__ void prepare_word(const Job &j, int64_t i) {
  const int ccon = j.transpose ? j.B : j.A, ccols = j.transpose ? j.A : j.B;
  const int64_t total = (int64_t)j.K * j.nt16 * j.ns * 64;
  if (i >= total) return;
  const int lane = (int)(i & 63);
  int64_t b = i >> 6;
  const int s = (int)(b % j.ns); b /= j.ns;
  const int gt = (int)(b % j.nt16);
  const int k = (int)(b / j.nt16);
  const int n = lane & 15, g = lane >> 4;
  const int col = (gt / j.nctt) * 16 * j.nctt + h_local_col(j.nctt, gt % j.nctt, n);
  uint16_t h[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int c = 32 * s + 8 * g + q;
    float v = 0.f;
    if (col < ccols && c < ccon)
      v = j.transpose ? j.src[((int64_t)k * j.A + col) * j.B + c] : j.src[((int64_t)k * j.A + c) * j.B + col];
    h[q] = f2h(HT{}, v);
  }
  uint4 o;
  o.x = h[0] | ((uint32_t)h[1] << 16); o.y = h[2] | ((uint32_t)h[3] << 16);
  o.z = h[4] | ((uint32_t)h[5] << 16); o.w = h[6] | ((uint32_t)h[7] << 16);
  reinterpret_cast<uint4 *>(j.dst)[i] = o;
}

__global__ void __launch_bounds__(256) weights_multi_kernel(const Job *__restrict__ jobs, int njobs) {
  __shared__ float tile[32][33];
  // the job whose block range holds this workgroup: last job with first_block <= blockIdx.x
  int lo = 0, hi = njobs - 1;
  const int64_t wb = blockIdx.x;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].first_block <= wb) lo = mid; else hi = mid - 1;
  }
  const Job j = jobs[lo];
  const int64_t lb = wb - j.first_block;
  if (j.kind == 0) transpose_tile(j, lb, tile);
  else if (j.kind == 1) prepare_word<Bf16>(j, lb * 256 + threadIdx.x);
  else prepare_word<Fp16>(j, lb * 256 + threadIdx.x);
}

int64_t job_blocks(pcs_weight_job &j) {
  if (j.kind == 0) return (int64_t)j.K * ceil_div(j.A, 32) * ceil_div(j.B, 32);
  const int ccon = j.transpose ? j.B : j.A, ccols = j.transpose ? j.A : j.B;
  j.nctt = conv_nctt(ccols);
  j.nt16 = (int)ceil_div(ccols, 16 * j.nctt) * j.nctt;
  j.ns = (int)ceil_div(ccon, 32);
  return ceil_div((int64_t)j.K * j.nt16 * j.ns * 64, 256);
}

}  // namespace

extern "C" int64_t pcs_weights_multi_plan(pcs_weight_job *jobs_host, int32_t n_jobs) {
  if (n_jobs < 0 || (n_jobs > 0 && !jobs_host)) { set_error("pcs_weights_multi_plan: bad args"); return -1; }
  int64_t blocks = 0;
  for (int i = 0; i < n_jobs; ++i) {
    pcs_weight_job &j = jobs_host[i];
    if (j.K <= 0 || j.A <= 0 || j.B <= 0 || j.kind < 0 || j.kind > 2 || !j.src || !j.dst ||
        (j.kind != 0 && !convh_applies(j.transpose ? j.B : j.A, j.transpose ? j.A : j.B, j.K))) {
      set_error("pcs_weights_multi_plan: job %d: bad shape / kind / pointer (half kinds: shapes of pcs_conv_h_applies only)", i);
      return -1;
    }
    j.first_block = blocks;
    blocks += job_blocks(j);
  }
  if (blocks > 0x7FFFFFFF) { set_error("pcs_weights_multi_plan: too many work blocks"); return -1; }
  return blocks;
}

extern "C" int pcs_weights_multi(const pcs_weight_job *jobs_dev, int32_t n_jobs, int64_t total_blocks, void *stream) {
  if (n_jobs < 0 || total_blocks < 0 || total_blocks > 0x7FFFFFFF || (n_jobs > 0 && !jobs_dev)) {
    set_error("pcs_weights_multi: bad args");
    return PCS_EINVAL;
  }
  if (n_jobs == 0 || total_blocks == 0) return PCS_OK;
  hipLaunchKernelGGL(weights_multi_kernel, dim3((unsigned)total_blocks), dim3(256), 0, as_stream(stream),
                     reinterpret_cast<const Job *>(jobs_dev), (int)n_jobs);
  return check_launch("pcs_weights_multi");
}
