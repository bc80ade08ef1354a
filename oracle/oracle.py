"""NumPy-level oracle API over pcs_oracle.c + pure-NumPy restatements of the reference's
Python-side rulebook logic.

TEST INFRASTRUCTURE ONLY (see the header of pcs_oracle.c): imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg, never by openpcseg_amd/.
Every function cites the reference lines it restates (TS = package/torchsparse.zip,
prefix torchsparse/).
"""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(HERE, "pcs_oracle.c")
_SO = os.path.join(HERE, "libpcs_oracle.so")
_lib = None


def build(force=False):
    """gcc the C restatement (-O3 (SSE2 auto-vectorisation only: no FMA contraction, no reassociation), no OpenMP: cores = 1 when timed as a baseline)."""
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(_SRC):
        subprocess.check_call(["gcc", "-O3", "-std=c99", "-fPIC", "-shared", _SRC, "-o", _SO])
    return _SO


def _c():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


I64, I32 = ctypes.c_int64, ctypes.c_int32


# ---- hashing -----------------------------------------------------------------------------------
def sphash(coords, offsets=None):
    """TS:torchsparse/nn/functional/hash.py:10-37 over hash_cuda.cu:10-55."""
    coords = _i32(coords)
    n = coords.shape[0]
    if offsets is None:
        out = np.empty(n, dtype=np.int64)
        _c().orc_hash(_p(coords), I64(n), _p(out))
        return out
    offsets = _i32(offsets)
    k = offsets.shape[0]
    out = np.empty((k, n), dtype=np.int64)
    _c().orc_kernel_hash(_p(coords), I64(n), _p(offsets), I32(k), _p(out))
    return out


def sphashquery(queries, references):
    """TS:torchsparse/nn/functional/query.py:8-33 over query_cpu.cpp:12-37: index of the
    FIRST reference equal to each query, -1 when absent."""
    queries = np.asarray(queries, dtype=np.int64)
    references = np.asarray(references, dtype=np.int64)
    if references.size == 0:
        return np.full(queries.shape, -1, dtype=np.int64)
    uniq, first = np.unique(references, return_index=True)
    pos = np.searchsorted(uniq, queries.reshape(-1))
    pos = np.clip(pos, 0, len(uniq) - 1)
    hit = uniq[pos] == queries.reshape(-1)
    return np.where(hit, first[pos], -1).astype(np.int64).reshape(queries.shape)


def spcount(idx, num):
    """TS:torchsparse/backend/others/count_cuda.cu:10-16."""
    idx = np.asarray(idx)
    return np.bincount(idx[idx >= 0], minlength=int(num)).astype(np.int32)[: int(num)]


# ---- point <-> voxel -----------------------------------------------------------------------------
def voxelize_fwd(feats, idx, counts):
    feats, idx, counts = _f32(feats), _i32(idx), _i32(counts)
    n, c = feats.shape
    m = counts.shape[0]
    out = np.empty((m, c), dtype=np.float32)
    _c().orc_voxelize_fwd(_p(feats), _p(idx), _p(counts), I64(n), I64(m), I32(c), _p(out))
    return out


def voxelize_bwd(gout, idx, counts, n):
    gout, idx, counts = _f32(gout), _i32(idx), _i32(counts)
    c = gout.shape[1]
    gin = np.empty((n, c), dtype=np.float32)
    _c().orc_voxelize_bwd(_p(gout), _p(idx), _p(counts), I64(n), I32(c), _p(gin))
    return gin


def devoxelize_fwd(feat, idx8, w8):
    feat, idx8, w8 = _f32(feat), _i32(idx8), _f32(w8)
    n, c = idx8.shape[0], feat.shape[1]
    out = np.empty((n, c), dtype=np.float32)
    _c().orc_devoxelize_fwd(_p(feat), _p(idx8), _p(w8), I64(n), I32(c), _p(out))
    return out


def devoxelize_bwd(gout, idx8, w8, m):
    gout, idx8, w8 = _f32(gout), _i32(idx8), _f32(w8)
    n, c = gout.shape
    gfeat = np.empty((m, c), dtype=np.float32)
    _c().orc_devoxelize_bwd(_p(gout), _p(idx8), _p(w8), I64(n), I64(m), I32(c), _p(gfeat))
    return gfeat


def calc_ti_weights(coords, idx_query, scale=1):
    """TS:torchsparse/nn/functional/devoxelize.py:10-48, fp32 op for op. idx_query (8,N)."""
    p = np.asarray(coords, dtype=np.float32)[:, :3]
    s = np.float32(scale)
    pf = np.floor(p / s) * s if scale != 1 else np.floor(p)
    pc = pf + s
    x, y, z = p[:, 0], p[:, 1], p[:, 2]
    xf, yf, zf = pf[:, 0], pf[:, 1], pf[:, 2]
    xc, yc, zc = pc[:, 0], pc[:, 1], pc[:, 2]
    w = np.stack([
        (xc - x) * (yc - y) * (zc - z), (xc - x) * (yc - y) * (z - zf),
        (xc - x) * (y - yf) * (zc - z), (xc - x) * (y - yf) * (z - zf),
        (x - xf) * (yc - y) * (zc - z), (x - xf) * (yc - y) * (z - zf),
        (x - xf) * (y - yf) * (zc - z), (x - xf) * (y - yf) * (z - zf)], axis=0).astype(np.float32)
    if scale != 1:
        w = w / np.float32(scale ** 3)
    w[np.asarray(idx_query) == -1] = 0
    w = w / (w.sum(axis=0, dtype=np.float32) + np.float32(1e-8))
    return w.astype(np.float32)


# ---- kernel offsets / downsample / rulebook -----------------------------------------------------
def _ntuple(x):
    return tuple(x) if isinstance(x, (list, tuple)) else (x, x, x)


def get_kernel_offsets(size, stride=1, dilation=1):
    """TS:torchsparse/nn/utils/kernel.py:11-32."""
    size, stride, dilation = _ntuple(size), _ntuple(stride), _ntuple(dilation)
    ax = [np.arange(-size[k] // 2 + 1, size[k] // 2 + 1) * stride[k] * dilation[k] for k in range(3)]
    if int(np.prod(size)) % 2 == 1:
        offs = [[x, y, z] for z in ax[2] for y in ax[1] for x in ax[0]]
    else:
        offs = [[x, y, z] for x in ax[0] for y in ax[1] for z in ax[2]]
    return np.array(offs, dtype=np.int32)


def spdownsample(coords, stride=2, kernel_size=2, tensor_stride=1):
    """TS:torchsparse/nn/functional/downsample.py:11-52."""
    coords = _i32(coords)
    stride, kernel_size, tensor_stride = _ntuple(stride), _ntuple(kernel_size), _ntuple(tensor_stride)
    ss = np.array([stride[k] * tensor_stride[k] for k in range(3)], dtype=np.int32)[None, :]
    if all(stride[k] in [1, kernel_size[k]] for k in range(3)):
        out = coords.copy()
        # torch.div(int, int) is a true (fp32) division, then trunc, then * stride (:25-28)
        q = np.trunc(coords[:, :3].astype(np.float32) / ss.astype(np.float32))
        out[:, :3] = (q * ss.astype(np.float32)).astype(np.int32)
    else:
        offsets = get_kernel_offsets(kernel_size, tensor_stride)
        cmin = coords[:, :3].min(axis=0, keepdims=True)
        x = (coords[:, None, :3] + offsets[None, :, :]).reshape(-1, 3)
        b = np.repeat(coords[:, 3:], offsets.shape[0], axis=1).reshape(-1, 1)
        cand = np.concatenate([x, b], axis=1)
        mask = (np.mod(cand[:, :3], ss) == 0) & (cand[:, :3] >= cmin)
        out = cand[mask.all(axis=1)]
    out = out[:, [3, 0, 1, 2]]
    out = np.unique(out, axis=0)  # lexicographic over (b, x, y, z)  (:49-51)
    return np.ascontiguousarray(out[:, [1, 2, 3, 0]], dtype=np.int32)


def build_kmap(in_coords, out_coords, kernel_size, in_stride=1, dilation=1, offsets=None):
    """TS:torchsparse/nn/functional/conv.py:156-176 -> (nbmaps (P,2) int64 [in,out], nbsizes (K,) int64)."""
    if offsets is None:
        offsets = get_kernel_offsets(kernel_size, in_stride, dilation)
    references = sphash(in_coords)
    queries = sphash(out_coords, offsets)
    results = sphashquery(queries, references)  # (K, N_out)
    nbsizes = (results != -1).sum(axis=1).astype(np.int64)
    kk, jj = np.nonzero(results != -1)  # row-major: k-major, out index ascending
    nbmaps = np.stack([results[kk, jj], jj], axis=1).astype(np.int64)
    return nbmaps, nbsizes


# ---- convolution -----------------------------------------------------------------------------------
def conv_fwd(feats, weight, nbmaps, nbsizes, sizes, transposed=False):
    """TS:torchsparse/nn/functional/conv.py:16-82 -> out (sizes[1] or sizes[0], Cout)."""
    feats, weight = _f32(feats), _f32(weight)
    if weight.ndim == 2:
        weight = weight[None]
    nbm, nbs = _i32(nbmaps), _i32(nbsizes)
    k, cin, cout = weight.shape
    assert feats.shape[1] == cin, "Input feature size and kernel size mismatch"
    n_out = sizes[0] if transposed else sizes[1]
    out = np.empty((n_out, cout), dtype=np.float32)
    _c().orc_conv_fwd(_p(feats), _p(out), _p(weight), _p(nbm), _p(nbs), I64(n_out), I32(cin), I32(cout),
                      I32(k), I32(1 if transposed else 0))
    return out


def conv_bwd(feats, gout, weight, nbmaps, nbsizes, transposed=False):
    """TS:torchsparse/nn/functional/conv.py:84-119 -> (grad_input, grad_weight)."""
    feats, gout, weight = _f32(feats), _f32(gout), _f32(weight)
    wshape = weight.shape
    if weight.ndim == 2:
        weight = weight[None]
    nbm, nbs = _i32(nbmaps), _i32(nbsizes)
    k, cin, cout = weight.shape
    gin = np.empty_like(feats)
    gw = np.empty_like(weight)
    _c().orc_conv_bwd(_p(feats), _p(gin), _p(gout), _p(weight), _p(gw), _p(nbm), _p(nbs),
                      I64(feats.shape[0]), I32(cin), I32(cout), I32(k), I32(1 if transposed else 0))
    return gin, gw.reshape(wshape)


# ---- cylinder scatter (torch_scatter semantics restated; PARITY UNPINNED) ----------------------------
def scatter_max(src, index, m):
    """torch_scatter.scatter_max(src, index, dim=0) as used at R:tools/utils/common/seg_utils.py:178:
    per-voxel channel-wise max and the (first) arg-max point; empty voxel -> 0 / -1."""
    src = _f32(src)
    index = np.asarray(index, dtype=np.int64)
    n, c = src.shape
    out = np.zeros((m, c), dtype=np.float32)
    arg = np.full((m, c), -1, dtype=np.int64)
    for i in np.argsort(index, kind="stable"):
        v = index[i]
        upd = (arg[v] < 0) | (src[i] > out[v])
        out[v][upd] = src[i][upd]
        arg[v][upd] = i
    return out, arg


def scatter_max_bwd(gout, arg, n):
    gout = _f32(gout)
    m, c = gout.shape
    g = np.zeros((n, c), dtype=np.float32)
    vv, jj = np.nonzero(arg >= 0)
    g[arg[vv, jj], jj] = gout[vv, jj]
    return g


# ---- range_lib (RL = R:pcseg/model/segmentor/fusion/rpvnet/range_lib/) ------------------------------
# PINNED [r5]: oracle/build_ref_rangelib.py executes the reference's own kernel text (RL:range_utils/src/*.cu) on the host;
# tests/test_scatter_range.py compares these restatements (and the HIP kernels) with it. Rows outside the image are handled here
# with upper-bound checks the reference lacks (undefined behaviour there): not part of the pinned domain.
def map_count(pxpy, b, h, w):
    """RL:range_utils/src/map_count_gpu.cu:5-14 (+ upper-bound checks the reference lacks)."""
    pxpy = np.asarray(pxpy, dtype=np.int64)
    out = np.zeros((b, h, w), dtype=np.int32)
    ok = (pxpy[:, 0] >= 0) & (pxpy[:, 0] < b) & (pxpy[:, 1] >= 0) & (pxpy[:, 1] < w) & (pxpy[:, 2] >= 0) & (pxpy[:, 2] < h)
    np.add.at(out, (pxpy[ok, 0], pxpy[ok, 2], pxpy[ok, 1]), 1)
    return out


def denselize_fwd(feat, count_map, pxpy):
    """RL:range_utils/src/denselize_gpu.cu:5-19: out[b, :, py, px] += feat[i] / count[b, py, px]."""
    feat = _f32(feat)
    pxpy = np.asarray(pxpy, dtype=np.int64)
    b, h, w = count_map.shape
    out = np.zeros((b, feat.shape[1], h, w), dtype=np.float64)
    for i in range(feat.shape[0]):
        bb, px, py = pxpy[i]
        if 0 <= bb < b and 0 <= px < w and 0 <= py < h and count_map[bb, py, px] > 0:
            out[bb, :, py, px] += feat[i] / np.float32(count_map[bb, py, px])
    return out.astype(np.float32)


def denselize_bwd(gout, count_map, pxpy):
    """RL:range_utils/src/denselize_gpu.cu:21-34."""
    gout = _f32(gout)
    pxpy = np.asarray(pxpy, dtype=np.int64)
    b, c, h, w = gout.shape
    g = np.zeros((pxpy.shape[0], c), dtype=np.float32)
    for i in range(pxpy.shape[0]):
        bb, px, py = pxpy[i]
        if 0 <= bb < b and 0 <= px < w and 0 <= py < h and count_map[bb, py, px] > 0:
            g[i] = gout[bb, :, py, px] / np.float32(count_map[bb, py, px])
    return g


# ---- cylinder front-end ---------------------------------------------------------
def _range_corners(pxpy, h, w):
    """Corner pixels and bilinear weights of torch's grid_sampler_2d (mode='bilinear', padding_mode='zeros', align_corners=False;
    ATen GridSampler.h: grid_sampler_unnormalize + the nw / ne / sw / se weights), float32 like there -- what
    R:pcseg/model/segmentor/fusion/rpvnet/rpvnet.py:31-51 (`resample_grid_stacked`) calls per frame."""
    x, y = pxpy[:, 1].astype(np.float32), pxpy[:, 2].astype(np.float32)
    ix = ((x + np.float32(1)) * np.float32(w) - np.float32(1)) / np.float32(2)
    iy = ((y + np.float32(1)) * np.float32(h) - np.float32(1)) / np.float32(2)
    x0, y0 = np.floor(ix), np.floor(iy)
    x1, y1 = x0 + 1, y0 + 1
    wts = np.stack([(x1 - ix) * (y1 - iy), (ix - x0) * (y1 - iy), (x1 - ix) * (iy - y0), (ix - x0) * (iy - y0)], 1).astype(np.float32)
    cx = np.stack([x0, x1, x0, x1], 1).astype(np.int64)
    cy = np.stack([y0, y0, y1, y1], 1).astype(np.int64)
    ok = (cx >= 0) & (cx < w) & (cy >= 0) & (cy < h)
    return cx, cy, wts, ok


def range_sample_fwd(img, pxpy):
    """out (N, C): rows of frame f = pxpy[:, 0] sample img[f] (B, C, H, W) bilinearly; frames outside [0, B) give zeros (the
    reference's per-frame masks never select them). Accumulated corner by corner in float32 like the ATen kernel."""
    b, c, h, w = img.shape
    n = pxpy.shape[0]
    cx, cy, wts, ok = _range_corners(pxpy, h, w)
    f = pxpy[:, 0]
    fi = f.astype(np.int64)
    fok = (f >= 0) & (fi < b) & (fi == f)
    out = np.zeros((n, c), dtype=np.float32)
    fi = np.where(fok, fi, 0)
    for k in range(4):
        m = ok[:, k] & fok
        v = img[fi[m], :, cy[m, k], cx[m, k]].astype(np.float32)
        out[m] = out[m] + v * wts[m, k][:, None]
    return out


def range_sample_bwd(gout, pxpy, shape):
    """d img of range_sample_fwd (float64 accumulation: the summation order of a scatter is the implementation's)."""
    b, c, h, w = shape
    cx, cy, wts, ok = _range_corners(pxpy, h, w)
    f = pxpy[:, 0]
    fi = f.astype(np.int64)
    fok = (f >= 0) & (fi < b) & (fi == f)
    gimg = np.zeros((b, h, w, c), dtype=np.float64)
    for k in range(4):
        m = ok[:, k] & fok
        np.add.at(gimg, (fi[m], cy[m, k], cx[m, k]), gout[m].astype(np.float64) * wts[m, k][:, None].astype(np.float64))
    return np.ascontiguousarray(gimg.transpose(0, 3, 1, 2)).astype(np.float32)


def cylinder_partition(points, space_min, space_max, grid_size):
    """R:pcseg/data/dataset/semantickitti/semantickitti_cylinder.py:19-22 (cart2polar) + :144-159 ->
    (xyz_pol (n,3) f32 [rho, phi_deg, z], point_coord (n,3) int64, point_feature (n, 8 + extras) f32)."""
    points = np.asarray(points, dtype=np.float32)
    xyz = points[:, :3]
    rho = np.sqrt(xyz[:, 0] ** 2 + xyz[:, 1] ** 2)
    phi = np.arctan2(xyz[:, 1], xyz[:, 0])
    xyz_pol = np.stack((rho, phi, xyz[:, 2]), axis=1)
    xyz_pol[:, 1] = xyz_pol[:, 1] / np.pi * 180.
    lo, hi, grid = np.array(space_min), np.array(space_max), np.array(grid_size)
    intervals = (hi - lo) / (grid - 1)
    point_coord = np.floor((np.clip(xyz_pol, lo, hi) - lo) / intervals).astype(np.int64)
    centres = (point_coord.astype(np.float32) + 0.5) * intervals + lo
    feat = np.concatenate([centres, xyz_pol, points[:, :2], points[:, 3:]], axis=1).astype(np.float32)
    return xyz_pol, point_coord, feat


def voxelize_with_label(point_coord, point_labels, num_classes, ignore=67):
    """semantickitti_cylinder.py:31-45 -> (voxel_coords, voxel_labels, inds, inverse_map); the Python loop of :35-37
    as one np.add.at."""
    vox, inds, inverse = sparse_quantize(np.asarray(point_coord, dtype=np.int32))
    counter = np.zeros((vox.shape[0], num_classes), dtype=np.int64)
    labels = np.asarray(point_labels).reshape(-1)
    keep = labels != ignore
    np.add.at(counter, (inverse[keep], labels[keep]), 1)
    return vox, np.argmax(counter, axis=1), inds, inverse


def sparse_quantize(points, voxel_size=(1, 1, 1)):
    """TS:torchsparse/utils/quantize.py:9-46 restated without np.unique: voxel = floor(point / voxel_size) in float64
    (NumPy promotes the float32 points against the float64 voxel-size array, :37), key = row-major index inside
    the bounding box (ravel_hash :9-21), one representative per key = its first row, output ordered by key.
    Returns (vox (m,3) int32, index (m) int64, inverse (n) int64)."""
    pts = np.asarray(points)[:, :3].astype(np.float64)
    c = np.floor(pts / np.asarray(voxel_size, dtype=np.float64)[None, :]).astype(np.int32)
    rel = (c - c.min(axis=0)).astype(np.int64)
    ext = rel.max(axis=0) + 1
    key = (rel[:, 0] * ext[1] + rel[:, 1]) * ext[2] + rel[:, 2]
    order = np.lexsort((np.arange(key.size), key))  # by key, ties by row
    sk = key[order]
    head = np.ones(sk.size, dtype=bool)
    head[1:] = sk[1:] != sk[:-1]
    index = order[head].astype(np.int64)
    inverse = np.empty(key.size, dtype=np.int64)
    inverse[order] = np.cumsum(head) - 1
    return c[index], index, inverse


def lovasz_softmax(probas, labels, ignore=None):
    """Lovasz-softmax, classes='present', per_image=False, with d loss / d probas: the errors in float32 as the reference
    forms them (their ORDER is part of the function: 1 - p rounds in float32), everything after the sort in float64
    (R:tools/utils/common/lovasz_losses.py:158-204: flatten_probas drops the ignored points :207-228, then per present
    class errors = |fg - p_c|, descending sort, dot with lovasz_grad :23-35 of the sorted foreground; mean over the
    classes). The sort is STABLE (ties keep point order), which is what torch's radix sort gives the reference on the
    device. Labels outside [0, C) other than `ignore` are dropped like ignored ones (the reference would index-error
    or silently never match them). -> (loss float, grad (n, C) float64)."""
    p = np.asarray(probas, dtype=np.float32)
    lab = np.asarray(labels).astype(np.int64)
    n, nc = p.shape
    valid = (lab >= 0) & (lab < nc)
    if ignore is not None:
        valid &= lab != ignore
    grad = np.zeros((n, nc), np.float64)
    rows = np.nonzero(valid)[0]
    pv, lv = p[rows], lab[rows]
    losses = []
    for c in range(nc):
        fg = (lv == c).astype(np.float64)
        if fg.sum() == 0:
            continue
        err = np.abs(fg.astype(np.float32) - pv[:, c]).astype(np.float64)
        perm = np.argsort(-err, kind="stable")
        fgs = fg[perm]
        gts = fgs.sum()
        inter = gts - np.cumsum(fgs)
        union = gts + np.cumsum(1.0 - fgs)
        jac = 1.0 - inter / union
        g = jac.copy()
        g[1:] = jac[1:] - jac[:-1]
        losses.append(float(np.dot(err[perm], g)))
        d = np.zeros(len(rows))
        d[perm] = g
        grad[rows, c] = d * np.sign(pv[:, c].astype(np.float64) - fg)   # d|fg - p| / dp = sign(p - fg), 0 at a zero error
    if not losses:
        return 0.0, grad
    return float(np.mean(losses)), grad / len(losses)
