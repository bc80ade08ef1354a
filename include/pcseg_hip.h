/*
 * pcseg_hip.h -- C ABI of libpcseg_hip.so, the MI355X (gfx950) sparse-voxel hot path.
 *
 * This is the drop-in boundary ("B-B" in SURVEY.md section 8b): every entry point below
 * replaces one function of the reference's native module `torchsparse.backend`
 * (TS = /root/reference/package/torchsparse.zip, member prefix torchsparse/), i.e. what the
 * reference's pybind table TS:torchsparse/backend/pybind_cuda.cpp:18-39 binds, plus the
 * rulebook construction that the reference does in Python on top of those functions
 * (TS:torchsparse/nn/functional/conv.py:156-176, downsample.py:11-52).
 *
 * Conventions
 *   - plain C: raw DEVICE pointers + sizes, no torch types, no allocation inside the library.
 *     Scratch memory is caller-provided ("ws"); its size comes from the matching *_ws_bytes().
 *   - every launch goes to the hipStream_t passed as `stream` (void* here so that plain C
 *     callers need no HIP headers). Nothing synchronises the device or the host.
 *   - return value: PCS_OK (0) or a negative PCS_E* code; pcs_last_error() gives the text.
 *   - all tensors are dense row-major; "coords" rows are int32 [x, y, z, batch]
 *     (TS:torchsparse/tensor.py:10-21).
 *   - dtype suffix _f32 = IEEE fp32 storage and fp32 MFMA / FMA arithmetic.
 */
#ifndef PCSEG_HIP_H_
#define PCSEG_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PCS_OK 0
#define PCS_EINVAL (-1)   /* bad argument (shape / alignment / null pointer)            */
#define PCS_EWORKSPACE (-2) /* caller workspace too small                                */
#define PCS_ELAUNCH (-3)  /* hipLaunch / hipMemsetAsync failed (pcs_last_error has text) */
#define PCS_EUNSUPPORTED (-4)

#define PCS_ABI_VERSION 11

int pcs_abi_version(void);
const char *pcs_last_error(void);

/* ------------------------------------------------------------------------------------------
 * K1  hash_cuda          TS:torchsparse/backend/hash/hash_cuda.cu:10-23,67-73
 * 64-bit FNV-1a over the four uint32 words of a coord row, folded to 60 bits. Bit-exact.
 * coords (n,4) int32 -> out (n,) int64
 */
int pcs_hash(const int32_t *coords, int64_t n, int64_t *out, void *stream);

/* K2  kernel_hash_cuda   TS:torchsparse/backend/hash/hash_cuda.cu:27-55,75-84
 * hash of (coords[:, :3] + offsets[k], coords[:, 3]) for every kernel offset.
 * coords (n,4) int32, offsets (K,3) int32 -> out (K,n) int64, k-major. Bit-exact.
 */
int pcs_kernel_hash(const int32_t *coords, int64_t n, const int32_t *offsets, int32_t K,
                    int64_t *out, void *stream);

/* ------------------------------------------------------------------------------------------
 * K3-K5  hash_query_cuda  TS:torchsparse/backend/others/query_cuda.cu:9-56
 *                         TS:torchsparse/backend/hashmap/hashmap_cuda.cu:8-127
 * The reference builds a 3-function cuckoo table with a host-driven rehash loop on every
 * call. Here: ONE open-addressing (linear probe) table of {key:int64, value:int32}, built
 * by a single kernel with 64-bit CAS, no host round trip. Value = position of the key in
 * `keys`; with duplicate keys the SMALLEST position wins (the reference's CPU twin keeps the
 * first insert, TS:torchsparse/backend/others/query_cpu.cpp:22-26).
 * Table storage is caller-owned: `capacity` slots (power of two, from pcs_hashtable_capacity)
 * laid out as  uint64 key[capacity] | int32 val[capacity]  (pcs_hashtable_bytes).
 */
int64_t pcs_hashtable_capacity(int64_t n);
size_t pcs_hashtable_bytes(int64_t capacity);
int pcs_hashtable_build(const int64_t *keys, int64_t n, void *table, int64_t capacity,
                        void *stream);
/* out[i] = (position of queries[i] in keys) + 1, or 0 when absent -- the reference's
 * backend convention (query_cuda.cu / hashmap_cuda.cu:104-127; Python subtracts 1,
 * TS:torchsparse/nn/functional/query.py:32). */
int pcs_hashtable_query(const void *table, int64_t capacity, const int64_t *queries,
                        int64_t n1, int64_t *out, void *stream);

/* K6  count_cuda  TS:torchsparse/backend/others/count_cuda.cu:10-31
 * out[idx[i]] += 1 for idx[i] >= 0; out (s,) int32 is zeroed inside. */
int pcs_count(const int32_t *idx, int64_t n, int32_t *out, int64_t s, void *stream);

/* ------------------------------------------------------------------------------------------
 * K7/K8  voxelize_forward_cuda / voxelize_backward_cuda
 *        TS:torchsparse/backend/voxelize/voxelize_cuda.cu:12-80
 * fwd: out[idx[i], :] += feats[i, :] / counts[idx[i]]   (divide BEFORE accumulating, as the
 *      reference does); out (m,c) is zeroed inside; rows with idx[i] < 0 are skipped.
 * bwd: gin[i, :] = gout[idx[i], :] / counts[idx[i]]      (0 where idx[i] < 0)
 */
int pcs_voxelize_fwd_f32(const float *feats, const int32_t *idx, const int32_t *counts,
                         int64_t n, int64_t m, int32_t c, float *out, void *stream);
/* The CSR the contention-free forms below run over [v11]: index (n,) int32 = target row of every entry (a voxel per point for
 * K7, a voxel per (point, corner) for K10; < 0 or >= m = no row) -> order (n,) int64 = entry positions sorted by target row,
 * equal rows in ascending position (stable: the summation order of every output row is fixed), entries without a row behind
 * the last row; rowptr (m+1,) int64 = start of each row's run. One radix sort over the bits a row index needs + one binary
 * search per row. ws / ws_bytes from pcs_index_csr_ws_bytes(n, m) (0 on bad sizes); never allocates. */
size_t pcs_index_csr_ws_bytes(int64_t n, int64_t m);
int pcs_index_csr_i32(const int32_t *index, int64_t n, int64_t m, int64_t *order, int64_t *rowptr, void *ws,
                      size_t ws_bytes, void *stream);
/* contention-free forward: order int64 = point rows sorted by voxel (points with idx < 0 outside [rowptr[0], rowptr[m])),
 * rowptr (m+1) int64; every out row written once, deterministic summation order */
int pcs_voxelize_fwd_csr_f32(const float *feats, const int64_t *order, const int64_t *rowptr,
                             const int32_t *counts, int64_t m, int32_t c, float *out, void *stream);
int pcs_voxelize_bwd_f32(const float *gout, const int32_t *idx, const int32_t *counts,
                         int64_t n, int32_t c, float *gin, void *stream);

/* K9/K10  devoxelize_forward_cuda / devoxelize_backward_cuda
 *         TS:torchsparse/backend/devoxelize/devoxelize_cuda.cu:11-98
 * fwd: out[i,:] = sum_{k<8} w[i,k] * feat[idx[i,k],:]  (idx -1 => 0), accumulated in
 *      registers in k = 0..7 order and written once.
 * bwd: gfeat[idx[i,k],:] += w[i,k] * gout[i,:]; gfeat (m,c) is zeroed inside.
 * idx (n,8) int32, w (n,8) fp32.
 */
int pcs_devoxelize_fwd_f32(const float *feat, const int32_t *idx8, const float *w8, int64_t n,
                           int32_t c, float *out, void *stream);
int pcs_devoxelize_bwd_f32(const float *gout, const int32_t *idx8, const float *w8, int64_t n,
                           int64_t m, int32_t c, float *gfeat, void *stream);

/* voxel_to_point map in one pass (R:pcseg/model/segmentor/voxel/minkunet/utils.py:69-105: floor, cat, kernel_hash
 * over the 8 cell corners, hashquery, calc_ti_weights, two transposes): coords (n, coord_ld >= 4) float = x,y,z,..,batch
 * in stride-1 voxel units, table = pcs_hashtable_build over the hashes of the level's voxel coordinates ->
 * idx8 (n,8) int32 (voxel row, -1 absent; corner order of get_kernel_offsets(2, stride)) and w8 (n,8) trilinear
 * weights, the layout pcs_devoxelize_fwd_f32 reads. */
int pcs_corner_map_f32(const float *coords, int32_t coord_ld, int64_t n, int32_t stride, const void *table,
                       int64_t capacity, int32_t *idx8, float *w8, void *stream);

/* K10 without atomics: the same sum as pcs_devoxelize_bwd_f32, as a per-voxel segmented
 * reduction. order (E,) int64 = flat positions i*8+k of the VALID (idx >= 0) entries of idx8,
 * sorted by voxel; rowptr (m+1,) int64 = start of each voxel's run in `order`. Every gfeat row
 * is written once, in `order` order (deterministic). The caller builds order/rowptr once per
 * idx8 (one sort) and reuses them for every backward through the same map. */
int pcs_devoxelize_bwd_csr_f32(const float *gout, const int64_t *order, const int64_t *rowptr,
                               const float *w8, int64_t m, int32_t c, float *gfeat, void *stream);

/* calc_ti_weights  TS:torchsparse/nn/functional/devoxelize.py:10-48  (about 25 small torch
 * kernels in the reference, one kernel here). coords (n, coord_ld) fp32 (first 3 columns
 * used), idx_query (8,n) int64 (-1 = miss), -> w (8,n) fp32: trilinear corner weights in
 * get_kernel_offsets(2) order, /scale^3 when scale != 1, zeroed at misses, renormalised by
 * (sum + 1e-8). */
int pcs_ti_weights_f32(const float *coords, int32_t coord_ld, const int64_t *idx_query,
                       int64_t n, float scale, float *w, void *stream);

/* ------------------------------------------------------------------------------------------
 * spdownsample  TS:torchsparse/nn/functional/downsample.py:11-52
 * Step 1 (this call): candidate output coordinates as packed sortable 64-bit keys
 *   key = (batch << 54) | ((x + 2^17) << 36) | ((y + 2^17) << 18) | (z + 2^17)
 * whose ascending order equals the reference's lexicographic order over [b, x, y, z]
 * (downsample.py:49-51). mode 0 = fast branch (downsample.py:25-28): one key per input row,
 * coordinate truncated toward zero to a multiple of sample_stride[d]. mode 1 = general
 * branch (downsample.py:29-45): n*K keys for coords+offsets[k]; rows that fail
 * `% sample_stride == 0` or `>= coords_min` get the key INT64_MAX (sorts last).
 * Step 2: pcs_sort_unique_i64 (v11; or any ascending sort + unique of the caller's). Step 3: pcs_downsample_unpack.
 * *err (device int32, caller-zeroed) is set to 1 if a coordinate does not fit the packing
 * (|x|,|y|,|z| >= 2^17 or batch outside [0, 511]).
 * sample_stride3 is a HOST pointer (three small positive ints); offsets / coords_min3 are
 * device pointers.
 */
int pcs_downsample_pack(const int32_t *coords, int64_t n, const int32_t *sample_stride3,
                        int32_t mode, const int32_t *offsets, int32_t K,
                        const int32_t *coords_min3, int64_t *keys, int32_t *err, void *stream);
int pcs_downsample_unpack(const int64_t *keys, int64_t m, int32_t *coords, void *stream);
/* Step 2 of spdownsample [v11]: out[0 .. m) = the distinct keys in ascending (signed) order -- what the reference's
 * `torch.unique(coords, dim=0)` (downsample.py:47-51) yields on the packed rows. info (device int64[3]) receives everything the
 * host has to read back in one record: info[0] = m, info[1] = the largest key (INT64_MIN when m = 0; INT64_MAX = the general
 * branch's "rejected candidate" sentinel is present and is the last key), info[2] = *err_flag (0 when err_flag is NULL).
 * out holds n keys; ws / ws_bytes from pcs_sort_unique_ws_bytes(n) (never allocates; PCS_EWORKSPACE when too small). */
size_t pcs_sort_unique_ws_bytes(int64_t n);
int pcs_sort_unique_i64(const int64_t *keys, int64_t n, int64_t *out, int64_t *info, const int32_t *err_flag, void *ws,
                        size_t ws_bytes, void *stream);

/* ------------------------------------------------------------------------------------------
 * Rulebook (kernel map)  TS:torchsparse/nn/functional/conv.py:156-176
 * For every kernel offset k and every query row j (ascending): look up
 * hash(qcoords[j,:3] + offsets[k], qcoords[j,3]) in the table built over hash(ref coords).
 * The reference materialises a (K,N) int64 hash matrix, a (K,N) int64 result matrix, then
 * sum + nonzero. Here the hash is computed in registers and probed immediately:
 *   pass 1  pcs_rulebook_probe : results (K,nq) int32 (ref row or -1) + per-k hit counts
 *   pass 2  pcs_rulebook_fill  : pairs (P,2) int32 = (ref_row, query_row), ordered k-major
 *           then query_row ascending -- exactly the reference's nbmaps order
 *           (conv.py:169-172) -- and koff (K+1) int32 prefix offsets of each k's slice.
 * The same two calls with (query = input coords, ref = output coords, negated offsets)
 * give the input-sorted map used by dgrad / transposed convs.
 * nbsizes (K,) int64 mirrors the reference's `nbsizes` (conv.py:168).
 */
size_t pcs_rulebook_ws_bytes(int64_t nq, int32_t K);
int pcs_rulebook_probe(const int32_t *qcoords, int64_t nq, const int32_t *offsets, int32_t K,
                       const void *table, int64_t capacity, int32_t *results,
                       int64_t *nbsizes, void *ws, size_t ws_bytes, int32_t symmetric, void *stream);
/* symmetric != 0: the caller guarantees a submanifold map -- qcoords are the rows the table was built over, K is odd
 * and offsets[K-1-k] == -offsets[k]. Only the first K/2 offsets are probed; a hit (j + offsets[k] -> i) is also
 * recorded as (i + offsets[K-1-k] -> j) and the centre offset is the identity. Same results, half the probes. */
int pcs_rulebook_fill(const int32_t *results, int64_t nq, int32_t K, const void *ws,
                      int32_t *pairs, int32_t *koff, void *stream);
/* Per (k, tile) segment table for the output-stationary convolution: for tile t of
 * `tile_rows` consecutive destination rows, seg[k*(ntiles+1)+t] is the first pair of offset
 * k (absolute index into pairs) whose destination row >= t*tile_rows.
 * dst_col selects the column of `pairs` that holds the destination row (must be the sorted
 * one, i.e. 1 for maps produced by pcs_rulebook_fill). */
int pcs_rulebook_tile_segments(const int32_t *pairs, const int32_t *koff, int32_t K,
                               int64_t n_dst, int32_t tile_rows, int32_t dst_col,
                               int32_t *seg, void *stream);
/* Heaviest-first order of the row tiles of a segment table (work of a tile = its 16-row MFMA blocks over all
 * offsets): order[i] = the tile the i-th workgroup slot of the fused convolution runs. Workgroups are dispatched in
 * index order, so the light tiles run last and the launch drains evenly (+4..8 % on the deep levels). The order among
 * equally heavy tiles is unspecified; results never depend on the order. (PCS_TILE_ORDER_XCD=1, measured and not the
 * default: the slots of one XCD (i % 8) walk one contiguous eighth of the tiles, heaviest first inside it.) */
int pcs_rulebook_tile_order(const int32_t *seg, int32_t K, int64_t ntiles, int32_t *order, void *stream);

/* ------------------------------------------------------------------------------------------
 * F1  convolution_forward_cuda   TS:torchsparse/backend/convolution/convolution_cuda.cu:53-165
 * F2  convolution_backward_cuda  TS:torchsparse/backend/convolution/convolution_cuda.cu:167-278
 * (gather_kernel :14-24, scatter_kernel :27-37, torch::mm_out per offset :149, :259-263)
 *
 * pcs_conv_gather_gemm_f32: output-stationary fused gather-GEMM-accumulate
 *      dst[d, :] = sum over pairs (s, d) of offset k:  src[s, :] @ W[k]      (+ bias)
 * One workgroup owns `tile_rows` consecutive dst rows x a column tile, walks the kernel
 * offsets, gathers the needed src rows into LDS, contracts with W[k] on the fp32 MFMA pipe
 * and accumulates in LDS; every dst row is written exactly once (no atomics, no zero fill
 * needed, deterministic). Used for forward (src = input feats), for dgrad (src = grad_out,
 * W = per-offset transposed weights, input-sorted map) and for transposed convolutions.
 *   src (n_src, cin), W (K, cin, cout), dst (n_dst, cout); pairs (P,2) with the src row in
 *   column src_col and the dst row in column 1-src_col, sorted k-major / dst ascending;
 *   seg from pcs_rulebook_tile_segments with the same tile_rows; bias (cout) or NULL.
 *   tile_rows: a multiple of 16 in [16, 512]; shapes outside the wave kernels (cin % 4 == 0, cin >= 32, cout % 4 == 0,
 *   an even number of 16-column tiles, K <= 32 -- cin % 32 == 0 on the straight-line instance, other widths such as
 *   56 / 112 / 168 / 336 of RPVNet cr 1.75 on the TAIL instance) take 64 or 128 only (PCS_EUNSUPPORTED otherwise).
 * pcs_conv_tile_rows returns the default tile height for (cin, cout); pcs_conv_pick_tile_rows the
 * height for one layer call: with few dst rows (deep strides) the launch is only a few waves of
 * workgroups over the CUs, and the height is chosen so that the last wave is full.
 */
const char *pcs_conv_kernel_revision(void); /* identifies the fused-conv kernels a measurement file was taken on */
int32_t pcs_conv_tile_rows(int32_t cin, int32_t cout);
int32_t pcs_conv_pick_tile_rows(int64_t n_dst, int64_t n_pairs, int32_t K, int32_t cin, int32_t cout);
int32_t pcs_conv_pick_tile_rows_dt(int64_t n_dst, int64_t n_pairs, int32_t K, int32_t cin, int32_t cout,
                                   int32_t dtype); /* dtype 0: fp32 kernels (= the call above), 1 / 2: half kernels */
/* *   bn_partial (may be NULL): [ceil(n_dst / tile_rows)][2][cout] doubles. When given, the write-back also leaves, per
 *   tile and column, sum(x) and sum(x^2) of the rows it stored: the statistics pass of the BatchNorm that follows
 *   the convolution (R:pcseg/model/segmentor/voxel/minkunet/minkunet.py:31-129) costs no extra read of the tensor;
 *   pcs_bn_reduce_partials turns them into the `sums` vector of pcs_bn_finalize_f32. Only for shapes / tile heights
 *   with pcs_conv_emits_bn_partials(...) != 0 (PCS_EUNSUPPORTED otherwise).
 *   tile_order (may be NULL): ceil(n_dst / tile_rows) int32 from pcs_rulebook_tile_order for the same seg; NULL =
 *   tiles in row order (XCD-contiguous ranges). Only the wave kernels (16-byte-granular shapes from 32 input channels
 *   up; every shape of the half kernels) use it.
 */
int32_t pcs_conv_emits_bn_partials(int32_t cin, int32_t cout, int32_t K, int32_t tile_rows, int32_t dtype);
int32_t pcs_conv_uses_tile_order(int32_t cin, int32_t cout, int32_t K, int32_t dtype); /* 1: the shape's kernel reads tile_order */
int pcs_conv_gather_gemm_f32(const float *src, int64_t n_src, int32_t cin, const float *W,
                             int32_t K, int32_t cout, const int32_t *pairs, int32_t src_col,
                             const int32_t *seg, int32_t tile_rows, int64_t n_dst,
                             const float *bias, float *dst, double *bn_partial, const int32_t *tile_order,
                             void *stream);
/* Write-back extras of the fused convolution (all optional; the plain entry = all NULL). Used by the backward pass:
 *   addend : (n_dst, cout) in dst's dtype, added to every output row (dst = conv + bias + addend, in fp32 before any rounding).
 *            dgrad of a convolution whose input also feeds a residual / skip path
 *            (R:pcseg/model/segmentor/voxel/minkunet/minkunet.py:88-129 `relu(net(x) + downsample(x))`): the gradient the skip path
 *            hands back rides in, instead of a separate elementwise sum of the two gradients (what autograd does around
 *            TS:torchsparse/backend/convolution/convolution_cuda.cu:167-278).
 *   bn_x / bn_mask / bn_stat : this launch is the dgrad that produces dy of a BatchNorm (+ ReLU) OUTPUT (minkunet.py:31-129:
 *            conv -> BatchNorm -> ReLU -> conv): bn_x = that BatchNorm's input rows (n_dst, cout) in dst's dtype, bn_mask = its ReLU
 *            gate, one bit per element ((n_dst, cout / 32) words as pcs_bn_apply_* writes them; NULL = no ReLU), bn_stat = its
 *            mean[cout] | invstd[cout]. The write-back then leaves the BatchNorm's BACKWARD statistics sum(g), sum(g xhat),
 *            g = dy [y > 0], per tile in `bn_partial` ([tiles][2][cout] doubles; reduce with pcs_bn_bwd_reduce_partials) -- the
 *            statistics pass of torch's batch_norm_backward over (dy, x) disappears. Needs pcs_conv_emits_bn_partials() == 1.
 * Shapes the wave kernels do not serve (pcs_conv_supports_epilogue() == 0) return PCS_EUNSUPPORTED when any extra is set. */
typedef struct pcs_conv_epilogue {
  const void *addend;
  const void *bn_x;
  const uint32_t *bn_mask;
  const double *bn_stat;
  float act_slope;   /* LeakyReLU fused into the write-back: dst = v < 0 ? v * act_slope : v, applied before the store and before the
                      * forward BatchNorm statistics (R:pcseg/model/segmentor/voxel/cylinder3d/cylinder_ts.py:88-190: conv -> LeakyReLU
                      * -> BatchNorm1d). 0 is read as 1 (no activation), so a zero-initialised struct is the plain call. */
  int32_t reserved;
} pcs_conv_epilogue;
int pcs_conv_gather_gemm_f32_ex(const float *src, int64_t n_src, int32_t cin, const float *W,
                                int32_t K, int32_t cout, const int32_t *pairs, int32_t src_col,
                                const int32_t *seg, int32_t tile_rows, int64_t n_dst,
                                const float *bias, const pcs_conv_epilogue *ep, float *dst, double *bn_partial,
                                const int32_t *tile_order, void *stream);
int32_t pcs_conv_supports_epilogue(int32_t cin, int32_t cout, int32_t K, int32_t dtype);   /* dtype 0 fp32, 1 / 2 half kernels */

/* dst[k][b][a] = src[k][a][b]: the per-offset transposed weights dgrad contracts with (the reference transposes
 * inside torch::mm_out per offset, convolution_cuda.cu:259-263). */
int pcs_transpose_kab_f32(const float *src, int32_t K, int32_t A, int32_t B, float *dst, void *stream);

/* wgrad:  gW[k] = sum over pairs (a, b) of offset k:  fa[a, :]^T (outer) fb[b, :]
 *   fa (na, ca) rows indexed by pairs column a_col, fb (nb, cb) by the other column;
 *   gW (K, ca, cb). Deterministic two-pass split reduction; ws from pcs_conv_wgrad_ws_bytes.
 *   koff_dev / koff_host: the K+1 prefix offsets of each offset's slice of `pairs`, on the
 *   device (read by the kernels) and on the HOST (the caller has them from the rulebook
 *   build; they size the launch and the workspace). */
size_t pcs_conv_wgrad_ws_bytes(const int32_t *koff_host, int32_t K, int32_t ca, int32_t cb);
int pcs_conv_wgrad_f32(const float *fa, int32_t ca, const float *fb, int32_t cb,
                       const int32_t *pairs, int32_t a_col, const int32_t *koff_dev,
                       const int32_t *koff_host, int32_t K, float *gW, void *ws,
                       size_t ws_bytes, void *stream);

/* ------------------------------------------------------------------------------------------
 * Cylindrical scatter: torch_scatter.scatter_max(src, index, dim=0) as called by the reference's
 * cylinder front-end (R:tools/utils/common/seg_utils.py:178,
 * R:pcseg/model/segmentor/voxel/cylinder3d/cylinder_ts.py:35). torch_scatter is a third-party,
 * version-unpinned dependency that is not under /root/reference: PARITY UNPINNED (semantics
 * restated: per-voxel channel-wise max, argmax saved for backward, empty voxel -> 0 / -1).
 * Segmented form: order (E,) int64 = point ids sorted by voxel, rowptr (m+1,) int64.
 * (scatter_mean of the same front-end is pcs_voxelize_fwd_f32 with the voxel counts.)
 */
int pcs_scatter_max_fwd_f32(const float *src, const int64_t *order, const int64_t *rowptr, int64_t m,
                            int32_t c, float *out, int32_t *arg, void *stream);
/* gsrc (n,c) is zeroed inside, then gsrc[arg[v,j], j] = gout[v,j]. arg is int32 (point rows < 2^31): as int64 it
 * would be half of the forward kernel's HBM traffic; the Python layer widens it only if the caller reads it. */
int pcs_scatter_max_bwd_f32(const float *gout, const int32_t *arg, int64_t m, int64_t n, int32_t c,
                            float *gsrc, void *stream);

/* K13  map_count_forward   RL:range_utils/src/map_count_gpu.cu:5-14 (RL = R:pcseg/model/segmentor/
 *      fusion/rpvnet/range_lib/): out[b,py,px] += 1 for pxpy rows (b,px,py) inside the image
 *      (the reference checks only px,py >= 0); out (B,H,W) int32 zeroed inside.
 * K14  denselize_forward   RL:range_utils/src/denselize_gpu.cu:5-19: out[b,j,py,px] += feat[i,j] /
 *      count[b,py,px]; out (B,C,H,W) fp32 zeroed inside.
 * K15  denselize_backward  RL:range_utils/src/denselize_gpu.cu:21-34: gfeat[i,j] = gout[b,j,py,px] /
 *      count (0 where the reference would divide by zero / read out of bounds). */
int pcs_map_count(const int32_t *pxpy, int64_t n, int32_t B, int32_t H, int32_t W, int32_t *out,
                  void *stream);
int pcs_denselize_fwd_f32(const float *feat, const int32_t *count_map, const int32_t *pxpy, int64_t n,
                          int32_t B, int32_t C, int32_t H, int32_t W, float *out, void *stream);
/* contention-free forms (C % 4 == 0): order int64 = point rows sorted by pixel ((b*H + py)*W + px; out-of-range
 * points first), rowptr (B*H*W + 1) int64. forward: every out element written once (no memset), NCHW stores in full
 * lines, deterministic. backward: gout read in full lines, one 16-byte-vector row store per point (gfeat zeroed inside). */
int pcs_denselize_fwd_csr_f32(const float *feat, const int64_t *order, const int64_t *rowptr,
                              const int32_t *count_map, int32_t B, int32_t C, int32_t H, int32_t W,
                              float *out, void *stream);
int pcs_denselize_bwd_csr_f32(const float *gout, const int64_t *order, const int64_t *rowptr,
                              const int32_t *count_map, int64_t n, int32_t B, int32_t C, int32_t H, int32_t W,
                              float *gfeat, void *stream);
int pcs_denselize_bwd_f32(const float *gout, const int32_t *count_map, const int32_t *pxpy, int64_t n,
                          int32_t B, int32_t C, int32_t H, int32_t W, float *gfeat, void *stream);

/* ---- range image -> points (ABI v8) --------------------------------------------------------------------------------------
 * Replaces R:pcseg/model/segmentor/fusion/rpvnet/rpvnet.py:31-51 (`resample_grid_stacked` / `range_to_point`: a python loop over
 * frames around torch.nn.functional.grid_sample(mode='bilinear', padding_mode='zeros', align_corners=False)) for all frames in
 * one launch, and its backward (torch's grid_sampler_2d_backward: per-point channel loops of float atomics into NCHW planes,
 * 25 % of an RPVNet step) by an atomic-free segmented pass.
 *   img  (B, C, H, W) fp32;  pxpy (n, 3) fp32 rows (frame, x, y), x / y in [-1, 1];  out (n, C): out[p] = bilinear sample of
 *   frame pxpy[p][0] at ix = ((x + 1) W - 1) / 2, iy = ((y + 1) H - 1) / 2; corners outside the image contribute 0; a row whose
 *   frame is not an integer in [0, B) gives zeros. C % 4 == 0 (16-byte row pieces), else PCS_EUNSUPPORTED.
 *   backward: pcs_range_sample_corners writes, per point and corner (nw, ne, sw, se), the pixel key (b H + y) W + x (or -1) and the
 *   bilinear weight: keys / wts of 4 n entries. The caller sorts the keys (stable) into a CSR over the B H W pixels -- order
 *   (4 n entry ids, the -1 keys first) and rowptr (B H W + 1), once per pxpy and resolution -- and pcs_range_sample_bwd_csr_f32
 *   writes EVERY element of gimg (B, C, H, W): gimg[b, c, y, x] = sum over the pixel's entries of wts[e] * gout[e / 4, c], in
 *   ascending entry order (deterministic); no memset needed. */
int pcs_range_sample_fwd_f32(const float *img, const float *pxpy, int64_t n, int32_t B, int32_t C, int32_t H, int32_t W,
                             float *out, void *stream);
int pcs_range_sample_corners(const float *pxpy, int64_t n, int32_t B, int32_t H, int32_t W, int64_t *keys, float *wts,
                             void *stream);
int pcs_range_sample_bwd_csr_f32(const float *gout, const int64_t *order, const int64_t *rowptr, const float *wts,
                                 int32_t B, int32_t C, int32_t H, int32_t W, float *gimg, void *stream);

/* ------------------------------------------------------------------------------------------
 * Block fusion above the op boundary: training-mode BatchNorm over (N,C)
 * voxel features fused with the residual add and the ReLU that follow it in the reference's
 * blocks (R:pcseg/model/segmentor/voxel/minkunet/minkunet.py:31-129: Conv3d -> (Sync)BatchNorm
 * -> ReLU, and relu(net(x) + downsample(x))). Statistics are two-level and fixed-order
 * (deterministic); `sums` (2c + 1 doubles: sum x | sum x^2 | row count n) is the vector a data-parallel
 * run all-reduces between the stats and the finalize call -- SyncBatchNorm semantics; the global count
 * then reaches finalize / bwd_apply as the DEVICE pointer `count_dev` (= sums + 2c; overrides the host
 * `count` when non-NULL), so no rank ever reads it back --, `sums2` (sum g | sum g*xhat) the one it
 * all-reduces in backward. The forward partial sums are taken about a pivot row and un-shifted in
 * double (E[x^2] - mean^2 on raw fp32 sums cancels when |mean| >> std).
 *   forward : pcs_bn_stats_f32 -> [all-reduce sums, count] -> pcs_bn_finalize_f32 (stat = mean | invstd,
 *             running stats updated with the unbiased variance like nn.BatchNorm1d)
 *             -> pcs_bn_apply_f32: y = act((x - mean) * invstd * w + b [+ res])
 *   backward: pcs_bn_bwd_stats_f32 (g = dy * [y > 0] when relu) -> [all-reduce sums2]
 *             -> pcs_bn_bwd_apply_f32: dx = (g - sum_g/N - xhat * sum_gxhat/N) * invstd * w, dres = g
 *             (dw = sums2[c:], db = sums2[:c]). ABI v7: `sums2` holds 2c doubles FOLLOWED BY the same 2c values as floats
 *             (3c doubles of storage): the parameter gradients in the parameters' dtype without a conversion launch.
 *             ABI v8: the caller states the size of `sums2` in doubles (`sums2_doubles` >= 3c, else PCS_EWORKSPACE) -- a v6 caller's
 *             2c-double buffer fails loudly instead of being overrun. (v8 also drops pcs_conv_ring_enable / pcs_conv_ring_applies:
 *             the experimental ring kernels left the product library, tools/experimental/.)
 *   single process, statistics from the convolution's write-back: pcs_bn_reduce_partials_finalize = pcs_bn_reduce_partials +
 *             pcs_bn_finalize_f32 (count = n) in one launch (ABI v7; `sums` may be NULL there).
 * partial_ws: pcs_bn_num_partials() * 2 * c floats.
 * mask (optional, c % 32 == 0): n * c/32 words written by the apply pass, bit = [y > 0]; handed to the two backward
 *   passes instead of y (then y may be NULL) -- the ReLU gate costs 1/32 of a tensor read instead of a whole one.
 */
int32_t pcs_bn_num_partials(void);
int pcs_bn_stats_f32(const float *x, int64_t n, int32_t c, float *partial_ws, double *sums, void *stream);
int pcs_bn_reduce_partials(const double *partial, int64_t nrows, int32_t c, int64_t n, double *sums, void *stream);
/* Backward twin: [sum g | sum g xhat] (2c doubles followed by the same values as 2c floats; sums2_doubles >= 3c) from the per-tile
 * partials of a dgrad write-back (pcs_conv_gather_gemm_*_ex with bn_x): replaces pcs_bn_bwd_stats_* for that BatchNorm. */
int pcs_bn_bwd_reduce_partials(const double *partial, int64_t nrows, int32_t c, double *sums2, int64_t sums2_doubles, void *stream);
int pcs_bn_reduce_partials_finalize(const double *partial, int64_t nrows, int32_t c, int64_t n, double eps, double momentum,
                                    float *running_mean, float *running_var, double *sums, double *stat, void *stream);
int pcs_bn_finalize_f32(const double *sums, double count, const double *count_dev, int32_t c, double eps,
                        double momentum, float *running_mean, float *running_var, double *stat, void *stream);
/* concat fusion (torchsparse.cat([bn_relu(up_conv(x)), skip]), TS:torchsparse/operators.py:10-17 as used by
 * R:pcseg/model/segmentor/voxel/minkunet/minkunet.py:404-416): apply may write y as the left c columns of an (n, ldy)
 * buffer (ldy = row stride in elements, 0 = c) and copy the skip tensor `tail` (n, ctail; may be NULL / 0) into the
 * columns right of it in the same launch; the backward passes then read dy with the row stride lddy (0 = c) straight
 * out of the gradient of that buffer. */
int pcs_bn_apply_f32(const float *x, const float *res, const double *stat, const float *w, const float *b,
                     int64_t n, int32_t c, int32_t relu, float *y, uint32_t *mask, int64_t ldy, const float *tail,
                     int32_t ctail, void *stream);
int pcs_bn_bwd_stats_f32(const float *dy, const float *x, const float *y, const uint32_t *mask, const double *stat,
                         int64_t n, int32_t c, int32_t relu, float *partial_ws, double *sums2, int64_t sums2_doubles,
                         int64_t lddy, void *stream);
int pcs_bn_bwd_apply_f32(const float *dy, const float *x, const float *y, const uint32_t *mask, const double *stat,
                         const double *sums2, double count, const double *count_dev, const float *w, int64_t n,
                         int32_t c, int32_t relu, float *dx, float *dres, int64_t lddy, void *stream);
/* the same four passes over bf16 (dtype 1) / fp16 (dtype 2) feature tensors (x, res, y, dy, dx, dres all in `dtype`):
 * the mixed-precision pipeline of the reference (`--amp`), where the convolutions hand on halfs. Statistics,
 * scale / shift and the arithmetic stay fp32 / double; rows need 8-byte alignment for the vector path. */
int pcs_bn_stats_h(const void *x, int64_t n, int32_t c, int32_t dtype, float *partial_ws, double *sums, void *stream);
int pcs_bn_apply_h(const void *x, const void *res, const double *stat, const float *w, const float *b, int64_t n,
                   int32_t c, int32_t relu, int32_t dtype, void *y, uint32_t *mask, int64_t ldy, const void *tail,
                   int32_t ctail, void *stream);
int pcs_bn_bwd_stats_h(const void *dy, const void *x, const void *y, const uint32_t *mask, const double *stat, int64_t n,
                       int32_t c, int32_t relu, int32_t dtype, float *partial_ws, double *sums2, int64_t sums2_doubles,
                       int64_t lddy, void *stream);
int pcs_bn_bwd_apply_h(const void *dy, const void *x, const void *y, const uint32_t *mask, const double *stat,
                       const double *sums2, double count, const double *count_dev, const float *w, int64_t n,
                       int32_t c, int32_t relu, int32_t dtype, void *dx, void *dres, int64_t lddy, void *stream);
/* pcs_bn_bwd_apply_{f32,h} (dtype 0 / 1 / 2) for a BatchNorm whose INPUT is a LeakyReLU output (conv -> LeakyReLU -> BatchNorm1d,
 * R:pcseg/model/segmentor/voxel/cylinder3d/cylinder_ts.py:88-190): dx leaves multiplied by (x > 0 ? 1 : in_slope), i.e. as the
 * gradient of the activation's input -- torch's leaky_relu_backward pass disappears. */
int pcs_bn_bwd_apply_act(const void *dy, const void *x, const void *y, const uint32_t *mask, const double *stat,
                         const double *sums2, double count, const double *count_dev, const float *w, int64_t n, int32_t c,
                         int32_t relu, int32_t dtype, float in_slope, void *dx, void *dres, int64_t lddy, void *stream);

/* ---- device-side sparse_quantize ---------------------------------------------------------------
 * Replaces the dataloader-side NumPy voxel dedup TS:torchsparse/utils/quantize.py:9-46
 * (ravel_hash :9-21, sparse_quantize :24-46; called from R:pcseg/data/dataset/semantickitti/
 * semantickitti_voxel.py:112-120) for scans that are already resident in HBM. Contract: voxel =
 * floor(point / voxel_size) evaluated in double like NumPy does, one representative row per voxel =
 * its FIRST occurrence, voxels ordered by ascending ravel hash (row-major index inside the bounding box).
 *   floor: points (n, row_stride >= 3) float32 (is_float = 1), float64 (2) or int32 (0); voxel_size3 = 3 HOST doubles;
 *          coords (n,3) int32 out; bbox = 6 DEVICE int32 {min xyz, max xyz}, preset by the caller to
 *          {INT32_MAX x3, INT32_MIN x3}.
 *   keys:  keys[i] = ((x - xmin) * ey + (y - ymin)) * ez + (z - zmin), int64.
 *   the caller sorts (keys, row) STABLY, then  flags: 1 at the head of every run of equal keys;
 *   the caller scans the flags inclusively (rank), then
 *   emit:  vox (m,3) int32, index (m) int64 = representative row (may be NULL), inverse (n) int64 = voxel of
 *          every row (may be NULL); m = rank[n-1].
 */
int pcs_quantize_floor(const void *points, int32_t is_float, int64_t n, int32_t row_stride,
                       const double *voxel_size3, int32_t *coords, int32_t *bbox, void *stream);
int pcs_quantize_keys(const int32_t *coords, int64_t n, const int32_t *bbox, int64_t *keys, void *stream);
/* The same for a whole collated batch (TS:torchsparse/utils/quantize.py:15-21 per frame + TS:torchsparse/utils/collate.py:11-32):
 * key = ((frame * ex + x - x0) * ey + y - y0) * ez + z - z0 over the BATCH's bounding box bbox = {x0, y0, z0, x1, y1, z1} (device,
 * int32[6]); frames int64 per row. Ascending key = frames in order, inside a frame the reference's ravel-hash order. */
int pcs_quantize_frame_keys(const int32_t *coords, const int64_t *frames, int64_t n, const int32_t *bbox, int64_t *keys, void *stream);
int pcs_quantize_flags(const int64_t *sorted_keys, int64_t n, int32_t *flags, void *stream);
int pcs_quantize_emit(const int32_t *flags, const int64_t *rank, const int64_t *perm, const int32_t *coords,
                      int64_t n, int32_t *vox, int64_t *index, int64_t *inverse, void *stream);

/* Unique keys + inverse map + CSR row pointers from a STABLY sorted key vector (flags from pcs_quantize_flags, rank =
 * their inclusive scan, perm = the sort's row permutation): uniq (m), inverse (n) = run index of every original row,
 * rowptr (m + 1) = first sorted position of every run (rowptr[m] = n). One pass for what initial_voxelize
 * (R:pcseg/model/segmentor/voxel/minkunet/utils.py:16-19) gets from torch.unique + sphashquery + spcount; the sorted
 * order doubles as the CSR of the segmented spvoxelize that follows. */
int pcs_unique_emit(const int32_t *flags, const int64_t *rank, const int64_t *perm, const int64_t *sorted_keys,
                    int64_t n, int64_t *uniq, int64_t *inverse, int64_t *rowptr, void *stream);

/* ---- fp32 convolution on the 16-bit MFMAs ("bf16x3", opt-in) ------------------------------------------
 * The same operator as pcs_conv_gather_gemm_f32 (TS:torchsparse/backend/convolution/convolution_cuda.cu:53-165 in fp32),
 * fp32 features in and out, with every operand split into three bf16 planes and six plane products accumulated in fp32:
 * fp32-grade results (not bit-identical to an fp32 FMA chain), for callers that select it explicitly. gfx950 runs
 * fp32-input MFMAs at 1/16 of the bf16 rate and has no TF32 path.
 *   applies : cin % 8 == 0, cin >= 32, cout % 4 == 0, cout >= 32, K <= 32.
 *   prepare : W (K, A, B) fp32 -> Wp: three planes in MFMA fragment order (transpose = 1: the dgrad weights);
 *             bytes = pcs_conv_prepared_weights_x3_bytes(K, contraction, columns).
 *   conv    : arguments as pcs_conv_gather_gemm_f32 with Wp instead of W; tile_rows as picked for the fp32 kernels.
 */
int pcs_conv_x3_applies(int32_t cin, int32_t cout, int32_t K);
int32_t pcs_conv_x3_column_tiles(int32_t cout);
int32_t pcs_conv_x3_emits_bn_partials(int32_t cin, int32_t cout, int32_t K, int32_t tile_rows);
size_t pcs_conv_prepared_weights_x3_bytes(int32_t K, int32_t ccon, int32_t ccols);
int pcs_conv_prepare_weights_x3(const float *W, int32_t K, int32_t A, int32_t B, int32_t transpose, void *Wp, void *stream);
int pcs_conv_gather_gemm_f32_bf16x3(const float *src, int64_t n_src, int32_t cin, const void *Wp, int32_t K, int32_t cout,
                                    const int32_t *pairs, int32_t src_col, const int32_t *seg, int32_t tile_rows,
                                    int64_t n_dst, const float *bias, float *dst, double *bn_partial,
                                    const int32_t *tile_order, void *stream);

/* ---- half-precision convolution (bf16 / fp16 storage, 16-bit MFMA, fp32 accumulate) --------------
 * The mixed-precision path of the reference: under `--amp` its ops cast their inputs to half
 * (TS:torchsparse/nn/functional/conv.py:19) and convolution_cuda.cu:61,120-127 runs gather / mm / scatter
 * in half. dtype: 1 = bfloat16, 2 = float16 (features, prepared weights and outputs share it).
 * Served shapes: pcs_conv_h_applies(cin, cout, K) != 0 (cin % 8 == 0, cin >= 32, cout % 4 == 0, an even number of
 * 16-column tiles, K <= 32; a last contraction step of 8 / 16 / 24 channels meets zero-padded weight fragments);
 * callers convert other shapes (4/5-channel stems) to fp32 and use the _f32 entries.
 *   prepare : W (K, A, B) fp32 master weights -> Wp, the weights in MFMA fragment order, converted to `dtype`.
 *             transpose = 0: forward (contraction over A = cin, columns B = cout); transpose = 1: dgrad
 *             (contraction over B, columns A). Wp bytes = pcs_conv_prepared_weights_bytes(K, contraction, columns).
 *   conv    : pcs_conv_gather_gemm_f32 with src / dst in halfs and Wp instead of W; bias stays fp32 and is added
 *             in fp32 before the single rounding of the output.
 *   wgrad   : pcs_conv_wgrad_f32 with half operands; accumulated and returned in fp32 (gW, ws as in _f32).
 */
int pcs_conv_h_applies(int32_t cin, int32_t cout, int32_t K);
size_t pcs_conv_prepared_weights_bytes(int32_t K, int32_t contraction, int32_t columns);
int pcs_conv_prepare_weights_h(const float *W, int32_t K, int32_t A, int32_t B, int32_t transpose, int32_t dtype,
                               void *Wp, void *stream);
int pcs_conv_gather_gemm_h(const void *src, int64_t n_src, int32_t cin, const void *Wp, int32_t K, int32_t cout,
                           const int32_t *pairs, int32_t src_col, const int32_t *seg, int32_t tile_rows,
                           int64_t n_dst, const float *bias, void *dst, int32_t dtype, double *bn_partial,
                           const int32_t *tile_order, void *stream);
/* with the write-back extras of pcs_conv_gather_gemm_f32_ex (addend / bn_x in the storage dtype) */
int pcs_conv_gather_gemm_h_ex(const void *src, int64_t n_src, int32_t cin, const void *Wp, int32_t K, int32_t cout,
                              const int32_t *pairs, int32_t src_col, const int32_t *seg, int32_t tile_rows,
                              int64_t n_dst, const float *bias, const pcs_conv_epilogue *ep, void *dst, int32_t dtype,
                              double *bn_partial, const int32_t *tile_order, void *stream);
/* fp32 operands through the bf16 MFMAs (three-plane split, six products, fp32-grade result); same arguments as _f32 */
int pcs_conv_wgrad_f32_bf16x3(const float *fa, int32_t ca, const float *fb, int32_t cb, const int32_t *pairs,
                              int32_t a_col, const int32_t *koff_dev, const int32_t *koff_host, int32_t K, float *gW,
                              void *ws, size_t ws_bytes, void *stream);
int pcs_conv_wgrad_h(const void *fa, int32_t ca, const void *fb, int32_t cb, const int32_t *pairs, int32_t a_col,
                     const int32_t *koff_dev, const int32_t *koff_host, int32_t K, float *gW, void *ws,
                     size_t ws_bytes, int32_t dtype, void *stream);

/* ---- Cylinder3D front-end on the device -------------------------------
 * Replaces, for scans already resident in HBM, the per-frame NumPy work of
 * R:pcseg/data/dataset/semantickitti/semantickitti_cylinder.py (cart2polar :19-22, cylindrical partition
 * :144-160, voxelize_with_label :31-45) and the eval-time voxel -> point mapping of
 * R:pcseg/model/segmentor/voxel/minkunet/minkunet.py:441-453.
 *   partition : points (n, row_stride >= 3) float32 [x, y, z, extras...]; space_min3 / space_max3 = 3 HOST doubles
 *               (CYLINDER_SPACE_MIN / _MAX: rho, phi in degrees, z), grid3 = 3 HOST ints (CYLINDER_GRID_SIZE).
 *               polar (n,3) float32 [rho, phi_deg, z] (may be NULL); coord (n,3) int32 cell indices
 *               = floor((clip(polar, min, max) - min) / ((max - min) / (grid - 1))) in float64 like NumPy;
 *               feat (n, 8 + row_stride - 3) float32 = [cell centre (3), polar (3), x, y, extras] (may be NULL).
 *   label vote: voxel_labels[v] = first arg-max over classes of #{points i: inverse[i] == v, labels[i] == class},
 *               labels equal to ignore_label (67 in the reference) are not counted; counter_ws = m * num_classes
 *               int32 (zeroed inside); bad_flag = 1 DEVICE int32, set to 1 if a counted label is outside
 *               [0, num_classes) (the reference raises IndexError there).
 *   argmax    : out[i] = first arg-max of logits[inverse ? inverse[i] : i] (m rows of c floats), -1 for an
 *               out-of-range row: `out[cur_inv].argmax(1)` of the reference without the (n, c) intermediate.
 */
int pcs_cylinder_partition_f32(const float *points, int64_t n, int32_t row_stride, const double *space_min3,
                               const double *space_max3, const int32_t *grid3, float *polar, int32_t *coord,
                               float *feat, void *stream);
int pcs_voxel_label_vote(const int64_t *inverse, const int64_t *labels, int64_t n, int64_t m, int32_t num_classes,
                         int64_t ignore_label, int32_t *counter_ws, int32_t *bad_flag, int64_t *voxel_labels,
                         void *stream);
int pcs_rows_argmax_gather_f32(const float *logits, int64_t m, int32_t c, const int64_t *inverse, int64_t n,
                               int64_t *out, void *stream);

/* ---- all layers' weight preparation in one launch (ABI v7) ---------------------------------------------------------
 * pcs_transpose_kab_f32 (kind 0: dst (K, B, A) fp32 = per-offset transposed weights for dgrad) and
 * pcs_conv_prepare_weights_h (kind 1 bf16 / 2 fp16: dst = prepared half weights of pcs_conv_prepared_weights_bytes bytes,
 * `transpose` as there) for a whole list of layers: the weights change once per optimizer step, so a training step
 * needs one launch instead of one per layer call (csrc/weights_multi.hip). Element-wise identical to the per-layer calls.
 *   pcs_weights_multi_plan : fills nctt / nt16 / ns / first_block of the HOST job table, returns the launch's work-block
 *                            count (-1 + pcs_last_error on a bad job);
 *   pcs_weights_multi      : jobs_dev = the planned table copied to the device. */
typedef struct {
  const float *src; /* (K, A, B) fp32 master weights */
  void *dst;
  int32_t K, A, B;
  int32_t kind;      /* 0 transpose, 1 prepare bf16, 2 prepare fp16 */
  int32_t transpose; /* kinds 1 / 2: 0 = forward (contract over A), 1 = dgrad (contract over B) */
  int32_t nctt, nt16, ns; /* filled by the plan call */
  int64_t first_block;    /* filled by the plan call */
} pcs_weight_job;
int64_t pcs_weights_multi_plan(pcs_weight_job *jobs_host, int32_t n_jobs);
int pcs_weights_multi(const pcs_weight_job *jobs_dev, int32_t n_jobs, int64_t total_blocks, void *stream);

/* ---- Lovasz-softmax of the training criterion (the timed step's loss tail) -----------------
 * lovasz_softmax(probas, labels, classes='present', per_image=False, ignore) of
 * R:tools/utils/common/lovasz_losses.py:158-204 (+ lovasz_grad :23-35, flatten_probas :207-228) as the reference's
 * criterion calls it (R:pcseg/loss/__init__.py:106-115), value AND gradient w.r.t. probas, every class in one radix
 * sort instead of a sort / gather / three scans per class (csrc/lovasz.hip).
 *   probas (n, num_class) float32 row-major, labels (n,) int64. has_ignore = 0: every label inside [0, num_class)
 *   counts; has_ignore = 1: points labelled `ignore` (any value, also outside the class range) are dropped. Labels
 *   outside [0, num_class) never count. loss = 1 DEVICE float (0 when no class is present); grad (n, num_class)
 *   float32 = d loss / d probas, every element written (may be NULL: value only).
 *   Ties between equal errors keep point order (what torch's stable descending sort gives the reference).
 *   ws: pcs_lovasz_workspace_bytes(n, num_class, has_ignore, ignore) bytes (-1 + pcs_last_error on bad sizes;
 *   28 B per point and class + the sort's temporaries). num_class <= 60, n * num_class < 2^32 - 1. */
int64_t pcs_lovasz_workspace_bytes(int64_t n, int32_t num_class, int32_t has_ignore, int64_t ignore);
int pcs_lovasz_softmax_f32(const float *probas, const int64_t *labels, int64_t n, int32_t num_class, int32_t has_ignore,
                           int64_t ignore, float *loss, float *grad, void *ws, int64_t ws_bytes, void *stream);

/* ---- measurement switches (NOT part of the drop-in contract: no reference function stands behind them; results never depend on
 * them; used by tools/convh_ws_ab.py and tools/wgrad_interleave_ab.py to A/B two kernel policies inside one process) -------------
 * pcs_debug_convh_ws: mode -1 = environment (PCS_CONVH_WS, default on), 0 = the 16-bit convolution on conv_os5h only, 1 = the
 *   weight-stationary kernel where the policy picks it, >= 2 = wherever an instance exists; `ru` unused; rs = 1: two-row-block
 *   sub-groups on the two shapes that have both instances.
 * pcs_debug_wgrad_interleave: launch order of the weight-gradient splits, -1 = default (2 for 16-bit operands, 0 for fp32),
 *   0 offset-major, 1 position-major, 2 position-major with every XCD on a contiguous eighth. */
void pcs_debug_convh_ws(int32_t mode, int32_t ru, int32_t rs);
void pcs_debug_wgrad_interleave(int32_t mode);

#ifdef __cplusplus
}
#endif
#endif /* PCSEG_HIP_H_ */
