// Microbenchmark: issue rate of the fp32-input MFMAs on gfx950 (cycles per instruction per SIMD).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void k16(float *out, int iters, float a, float b) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
__global__ void k32(float *out, int iters, float a, float b) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename F>
void run(const char *name, F launch, int waves_per_simd, int nacc, double flops_per_inst) {
  float *out; hipMalloc(&out, 256 * 16 * 1024 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  launch(out, 10, waves_per_simd);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  launch(out, iters, waves_per_simd);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double insts_per_simd = (double)iters * nacc * waves_per_simd;
  double tflops = insts_per_simd * 1024 * flops_per_inst / (ms * 1e-3) / 1e12;
  printf("%-28s waves/SIMD=%d nacc=%d: %.3f ms  %.1f TFLOP/s  (%.1f ns per inst per SIMD)\n", name, waves_per_simd, nacc, ms,
         tflops, ms * 1e6 / insts_per_simd);
  hipFree(out);
}
int main() {
  for (int w : {1, 2, 4}) {
    run("mfma_f32_16x16x4f32 nacc=1", [](float *o, int it, int w) { k16<1><<<256, 256 * w>>>(o, it, 1.f, 2.f); }, w, 1, 2048);
    run("mfma_f32_16x16x4f32 nacc=2", [](float *o, int it, int w) { k16<2><<<256, 256 * w>>>(o, it, 1.f, 2.f); }, w, 2, 2048);
    run("mfma_f32_16x16x4f32 nacc=6", [](float *o, int it, int w) { k16<6><<<256, 256 * w>>>(o, it, 1.f, 2.f); }, w, 6, 2048);
    run("mfma_f32_32x32x2f32 nacc=1", [](float *o, int it, int w) { k32<1><<<256, 256 * w>>>(o, it, 1.f, 2.f); }, w, 1, 4096);
    run("mfma_f32_32x32x2f32 nacc=2", [](float *o, int it, int w) { k32<2><<<256, 256 * w>>>(o, it, 1.f, 2.f); }, w, 2, 4096);
  }
  return 0;
}
