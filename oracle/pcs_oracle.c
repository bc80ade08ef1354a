/*
 * pcs_oracle.c -- CPU restatement of the reference's sparse-voxel hot path.
 *
 * TEST INFRASTRUCTURE ONLY. Nothing under openpcseg_amd/ may import, link or execute this
 * file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and
 * only as the checker. Plain scalar C, one loop nest per reference function, each citing the
 * reference lines it follows (TS = /root/reference/package/torchsparse.zip, prefix
 * torchsparse/). Pinned against the reference's own compiled CPU backend (oracle/_ref,
 * oracle/build_ref.py) and the four hash known-answers of SURVEY.md section 2.2 in
 * tests/test_oracle_pinning.py.
 *
 * Where the reference's CPU twin is wrong the CUDA source is the authority:
 *   - kernel_hash: hash_cpu.cpp:29 reads data[3] for every row; hash_cuda.cu:42-46 reads the
 *     row's own batch index  -> followed here.
 *   - devoxelize backward: devoxelize_cpu.cpp:51-53 indexes top_grad with the voxel index and
 *     writes through -1; devoxelize_cuda.cu:37-57 is followed here.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* TS:torchsparse/backend/hash/hash_cuda.cu:10-23 (== hash_cpu.cpp:7-18) */
static uint64_t fnv60(const int32_t c[4]) {
  uint64_t h = 14695981039346656037ULL;
  for (int j = 0; j < 4; j++) {
    h ^= (uint32_t)c[j];
    h *= 1099511628211ULL;
  }
  return (h >> 60) ^ (h & 0x0FFFFFFFFFFFFFFFULL);
}

void orc_hash(const int32_t *coords, int64_t n, int64_t *out) {
  for (int64_t i = 0; i < n; i++) out[i] = (int64_t)fnv60(coords + 4 * i);
}

/* TS:torchsparse/backend/hash/hash_cuda.cu:27-55 ; out is (K, n) k-major (:53) */
void orc_kernel_hash(const int32_t *coords, int64_t n, const int32_t *offsets, int32_t K,
                     int64_t *out) {
  for (int32_t k = 0; k < K; k++) {
    for (int64_t i = 0; i < n; i++) {
      int32_t c[4];
      for (int j = 0; j < 3; j++) c[j] = coords[4 * i + j] + offsets[3 * k + j];
      c[3] = coords[4 * i + 3];
      out[(int64_t)k * n + i] = (int64_t)fnv60(c);
    }
  }
}

/* TS:torchsparse/backend/voxelize/voxelize_cuda.cu:12-25 (== voxelize_cpu.cpp:7-25):
 * divide by the count BEFORE accumulating; points visited in index order. */
void orc_voxelize_fwd(const float *feats, const int32_t *idx, const int32_t *counts, int64_t n,
                      int64_t m, int32_t c, float *out) {
  memset(out, 0, (size_t)m * c * sizeof(float));
  for (int64_t i = 0; i < n; i++) {
    int32_t pos = idx[i];
    if (pos < 0 || counts[pos] == 0) continue; /* voxelize_cuda.cu:22 */
    float cnt = (float)counts[pos];
    for (int32_t j = 0; j < c; j++) out[(int64_t)pos * c + j] += feats[i * c + j] / cnt;
  }
}

/* TS:torchsparse/backend/voxelize/voxelize_cuda.cu:28-42 */
void orc_voxelize_bwd(const float *gout, const int32_t *idx, const int32_t *counts, int64_t n,
                      int32_t c, float *gin) {
  memset(gin, 0, (size_t)n * c * sizeof(float));
  for (int64_t i = 0; i < n; i++) {
    int32_t pos = idx[i];
    if (pos < 0 || counts[pos] == 0) continue;
    float cnt = (float)counts[pos];
    for (int32_t j = 0; j < c; j++) gin[i * c + j] = gout[(int64_t)pos * c + j] / cnt;
  }
}

/* TS:torchsparse/backend/devoxelize/devoxelize_cuda.cu:11-33: corners accumulated k = 0..7 */
void orc_devoxelize_fwd(const float *feat, const int32_t *idx8, const float *w8, int64_t n,
                        int32_t c, float *out) {
  for (int64_t i = 0; i < n; i++) {
    for (int32_t j = 0; j < c; j++) {
      float acc = 0.f;
      for (int k = 0; k < 8; k++) {
        int32_t id = idx8[i * 8 + k];
        float f = id >= 0 ? feat[(int64_t)id * c + j] : 0.f;
        acc += w8[i * 8 + k] * f;
      }
      out[i * c + j] = acc;
    }
  }
}

/* TS:torchsparse/backend/devoxelize/devoxelize_cuda.cu:37-57 (authority; the CPU twin is
 * broken). Accumulated in double so that the check does not depend on summation order. */
void orc_devoxelize_bwd(const float *gout, const int32_t *idx8, const float *w8, int64_t n,
                        int64_t m, int32_t c, float *gfeat) {
  double *acc = (double *)calloc((size_t)m * c, sizeof(double));
  for (int64_t i = 0; i < n; i++) {
    for (int k = 0; k < 8; k++) {
      int32_t id = idx8[i * 8 + k];
      if (id < 0) continue;
      float w = w8[i * 8 + k];
      for (int32_t j = 0; j < c; j++) acc[(int64_t)id * c + j] += (double)(w * gout[i * c + j]);
    }
  }
  for (int64_t e = 0; e < m * c; e++) gfeat[e] = (float)acc[e];
  free(acc);
}

/* TS:torchsparse/backend/convolution/convolution_cuda.cu:53-165 (== convolution_cpu.cpp:38-117)
 * for k: buf = gather(in, nbmaps[:, transpose]); buf2 = buf @ W[k]; out[nbmaps[:, 1-transpose]] += buf2
 * The centre-offset shortcut (:76-88) is the same arithmetic on a zeroed output.
 * nbmaps (P,2) int32 rows (in_idx, out_idx), k-major; nbsizes (K). */
void orc_conv_fwd(const float *in, float *out, const float *W, const int32_t *nbmaps,
                  const int32_t *nbsizes, int64_t n_out_rows, int32_t cin, int32_t cout,
                  int32_t K, int32_t transpose) {
  memset(out, 0, (size_t)n_out_rows * cout * sizeof(float));
  float *tmp = (float *)malloc((size_t)cout * sizeof(float));
  int64_t p = 0;
  for (int32_t k = 0; k < K; k++) {
    const float *Wk = W + (int64_t)k * cin * cout;
    for (int32_t q = 0; q < nbsizes[k]; q++, p++) {
      int32_t i = nbmaps[2 * p + transpose];
      int32_t o = nbmaps[2 * p + 1 - transpose];
      if (i < 0 || o < 0) continue;
      for (int32_t b = 0; b < cout; b++) tmp[b] = 0.f;
      for (int32_t a = 0; a < cin; a++) {
        float x = in[(int64_t)i * cin + a];
        const float *w = Wk + (int64_t)a * cout;
        for (int32_t b = 0; b < cout; b++) tmp[b] += x * w[b];
      }
      for (int32_t b = 0; b < cout; b++) out[(int64_t)o * cout + b] += tmp[b];
    }
  }
  free(tmp);
}

/* TS:torchsparse/backend/convolution/convolution_cuda.cu:167-278 (== convolution_cpu.cpp:119-183)
 * grad_in[in] += grad_out[out] @ W[k]^T ;  grad_W[k] += in[in]^T (x) grad_out[out]
 * (roles of the map columns as in :243-263; `transpose` swaps them). */
void orc_conv_bwd(const float *in, float *gin, const float *gout, const float *W, float *gW,
                  const int32_t *nbmaps, const int32_t *nbsizes, int64_t n_in_rows, int32_t cin,
                  int32_t cout, int32_t K, int32_t transpose) {
  memset(gin, 0, (size_t)n_in_rows * cin * sizeof(float));
  double *gw = (double *)calloc((size_t)K * cin * cout, sizeof(double));
  int64_t p = 0;
  for (int32_t k = 0; k < K; k++) {
    const float *Wk = W + (int64_t)k * cin * cout;
    double *gwk = gw + (int64_t)k * cin * cout;
    for (int32_t q = 0; q < nbsizes[k]; q++, p++) {
      int32_t i = nbmaps[2 * p + transpose];     /* row of `in` / grad_in   */
      int32_t o = nbmaps[2 * p + 1 - transpose]; /* row of grad_out         */
      if (i < 0 || o < 0) continue;
      const float *g = gout + (int64_t)o * cout;
      const float *x = in + (int64_t)i * cin;
      for (int32_t a = 0; a < cin; a++) {
        const float *w = Wk + (int64_t)a * cout;
        float s = 0.f;
        for (int32_t b = 0; b < cout; b++) {
          s += g[b] * w[b];
          gwk[(int64_t)a * cout + b] += (double)(x[a] * g[b]);
        }
        gin[(int64_t)i * cin + a] += s;
      }
    }
  }
  for (int64_t e = 0; e < (int64_t)K * cin * cout; e++) gW[e] = (float)gw[e];
  free(gw);
}
