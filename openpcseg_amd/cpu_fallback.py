"""Pure-PyTorch CPU path of the operator set (BASELINE config 1: "pure-PyTorch CPU scatter_add fallback, world_size=1").

The reference keeps a device-agnostic restatement of its convolution next to the native call
(TS:torchsparse/nn/functional/conv.py:67-79: per offset `output[o] += mm(input[i], W[k])`) and `_cpu` twins of every
backend function (TS:torchsparse/backend/pybind_cuda.cpp:18-39). This module is that path for this package: every backend
method of openpcseg_amd.native.HipBackend that the operator API needs, written with torch gather / index_add_ /
searchsorted / unique on CPU tensors. It exists so that the API can be exercised and timed without a GPU.

It is NOT a fallback for device tensors and is never selected implicitly: HIP tensors always go to libpcseg_hip.so and
fail loudly when it is missing (native.py). A caller opts in for CPU tensors with

    with openpcseg_amd.cpu_fallback.enabled():      # or cpu_fallback.install() / PCS_CPU_PATH=1
        model(batch_on_cpu)

Independent of oracle/ (test infrastructure): nothing here imports it; tests compare the two.
"""
import contextlib
import os

import torch

from . import native

_FNV_OFFSET = 14695981039346656037 - (1 << 64)  # as signed int64
_FNV_PRIME = 1099511628211
_MASK60 = 0x0FFFFFFFFFFFFFFF


def _cpu(t, name):
    if not isinstance(t, torch.Tensor) or t.is_cuda:
        raise RuntimeError("openpcseg_amd.cpu_fallback: `%s` must be a CPU tensor (device tensors belong to the HIP backend)" % name)
    return t


class TorchKernelMap:
    """Same surface as native.KernelMap for the operator layer (pairs k-major, dst ascending inside an offset)."""

    def __init__(self, pairs, sizes, n_src, n_dst):
        self.pairs = pairs.int().contiguous()
        self._pairs_raw = self.pairs
        self.nbsizes = torch.as_tensor(sizes, dtype=torch.int64)
        self.koff_host = [0]
        for s in self.nbsizes.tolist():
            self.koff_host.append(self.koff_host[-1] + int(s))
        self.koff = torch.tensor(self.koff_host, dtype=torch.int32)
        self.n_src, self.n_dst, self.K = n_src, n_dst, len(self.koff_host) - 1
        self.resolved = True

    @property
    def num_pairs(self):
        return self.koff_host[-1]

    def num_pairs_estimate(self):
        return self.koff_host[-1]

    def mirror(self):
        k, ko = self.K, self.koff_host
        pairs = torch.cat([self.pairs[ko[k - 1 - j]:ko[k - j]] for j in range(k)], dim=0)
        return TorchKernelMap(pairs, self.nbsizes.flip(0), self.n_dst, self.n_src)


class TorchCpuBackend:
    name = "torch-cpu"

    # -- K1 / K2: 60-bit FNV-1a over the unsigned 32-bit words x, y, z, batch (hash_cuda.cu:10-23); int64 arithmetic wraps
    @staticmethod
    def _fnv(cols):
        h = torch.full(cols[0].shape, _FNV_OFFSET, dtype=torch.int64)
        for c in cols:
            h = (h ^ (c.long() & 0xFFFFFFFF)) * _FNV_PRIME
        return ((h >> 60) & 0xF) ^ (h & _MASK60)

    def hash(self, coords):
        c = _cpu(coords, "coords")
        return self._fnv([c[:, 0], c[:, 1], c[:, 2], c[:, 3]])

    def kernel_hash(self, coords, offsets):
        c, o = _cpu(coords, "coords"), _cpu(offsets, "offsets")
        x = c[None, :, :3] + o[:, None, :]                      # (K, N, 3) int32: wraps like the reference's int adds
        b = c[None, :, 3].expand(o.shape[0], -1)
        return self._fnv([x[..., 0], x[..., 1], x[..., 2], b])

    # -- K3-K5: position of each query among the references, -1 when absent (first of equal references wins)
    def hash_query(self, queries, references):
        q, r = _cpu(queries, "queries").reshape(-1), _cpu(references, "references")
        if r.numel() == 0:
            return torch.full_like(q, -1)
        rs, order = torch.sort(r, stable=True)
        pos = torch.searchsorted(rs, q).clamp_(max=r.numel() - 1)
        hit = rs[pos] == q
        return torch.where(hit, order[pos], torch.full_like(q, -1))

    def count(self, idx, num):
        idx = _cpu(idx, "coords").long()
        return torch.bincount(idx[idx >= 0], minlength=int(num))[:int(num)].int()

    # -- K7-K10
    def voxelize_fwd(self, feats, idx, counts, cache_on=None):
        f, idx, cnt = _cpu(feats, "feats").float(), _cpu(idx, "coords").long(), _cpu(counts, "counts")
        ok = idx >= 0
        out = torch.zeros((cnt.shape[0], f.shape[1]), dtype=torch.float32)
        out.index_add_(0, idx[ok], f[ok] / cnt[idx[ok]].float().unsqueeze(1))  # K7 divides each addend (voxelize_cuda.cu:23)
        return out

    def voxelize_bwd(self, gout, idx, counts, n):
        g, idx, cnt = _cpu(gout, "grad_output").float(), _cpu(idx, "coords").long(), _cpu(counts, "counts")
        ok = idx >= 0
        gin = torch.zeros((n, g.shape[1]), dtype=torch.float32)
        gin[ok] = g[idx[ok]] / cnt[idx[ok]].float().unsqueeze(1)
        return gin

    def devoxelize_fwd(self, feats, idx8, w8):
        f, idx, w = _cpu(feats, "feats").float(), _cpu(idx8, "coords").long(), _cpu(w8, "weights").float()
        out = torch.zeros((idx.shape[0], f.shape[1]), dtype=torch.float32)
        for k in range(idx.shape[1]):
            ok = idx[:, k] >= 0
            out[ok] += w[ok, k:k + 1] * f[idx[ok, k]]
        return out

    def devoxelize_bwd(self, gout, idx8, w8, m):
        g, idx, w = _cpu(gout, "grad_output").float(), _cpu(idx8, "coords").long(), _cpu(w8, "weights").float()
        gf = torch.zeros((m, g.shape[1]), dtype=torch.float32)
        for k in range(idx.shape[1]):
            ok = idx[:, k] >= 0
            gf.index_add_(0, idx[ok, k], w[ok, k:k + 1] * g[ok])
        return gf

    def ti_weights(self, coords, idx_query, scale):
        """(8, N) trilinear weights, corner order z fastest (TS:torchsparse/nn/functional/devoxelize.py:10-48)."""
        p = _cpu(coords, "coords").float()[:, :3]
        pf = torch.floor(p / scale) * scale if scale != 1 else torch.floor(p)
        lo, hi = p - pf, pf + scale - p
        ws = []
        for dx in (0, 1):
            for dy in (0, 1):
                for dz in (0, 1):
                    ws.append((lo[:, 0] if dx else hi[:, 0]) * (lo[:, 1] if dy else hi[:, 1]) * (lo[:, 2] if dz else hi[:, 2]))
        w = torch.stack(ws, dim=0)
        if scale != 1:
            w = w / float(scale) ** 3
        w = torch.where(_cpu(idx_query, "idx_query") == -1, torch.zeros_like(w), w)
        return w / (w.sum(dim=0) + 1e-8)

    def level_table(self, voxel_coords):
        return self.hash(voxel_coords)

    def corner_map(self, point_coords, voxel_coords, stride):
        """(idx8 (N,8) int32, w8 (N,8)) of voxel_to_point (R:pcseg/model/segmentor/voxel/minkunet/utils.py:69-105)."""
        from .sparse import get_kernel_offsets
        pc = _cpu(point_coords, "coords").float()
        off = get_kernel_offsets(2, stride, 1, device="cpu")
        base = torch.cat([torch.floor(pc[:, :3] / stride) * stride, pc[:, 3:4]], dim=1).int()
        idx = self.hash_query(self.kernel_hash(base, off), self.hash(voxel_coords)).view(8, -1)
        w = self.ti_weights(pc, idx, stride)
        return idx.t().contiguous().int(), w.t().contiguous()

    # -- spdownsample (TS:torchsparse/nn/functional/downsample.py:25-51)
    def downsample(self, coords, sample_stride, offsets=None, coords_min=None):
        c = _cpu(coords, "coords").int()
        ss = torch.tensor([int(s) for s in sample_stride], dtype=torch.int32).unsqueeze(0)
        if offsets is None:
            c = c.clone()
            c[:, :3] = (torch.div(c[:, :3], ss).trunc() * ss).int()
        else:
            off = _cpu(offsets, "offsets").int()
            k = off.shape[0]
            x = c[:, :3].unsqueeze(1).repeat(1, k, 1) + off
            b = c[:, 3:].repeat(1, k)
            cand = torch.cat([x.view(-1, 3), b.view(-1, 1)], dim=1)
            mask = (cand[:, :3] % ss == 0) & (cand[:, :3] >= _cpu(coords_min, "coords_min").int().view(1, 3))
            c = cand[mask.all(dim=1)]
        c = torch.unique(c[:, [3, 0, 1, 2]], dim=0)
        return c[:, [1, 2, 3, 0]].contiguous()

    # -- rulebook (TS:torchsparse/nn/functional/conv.py:156-176): offset-major, output row ascending inside an offset
    def build_kmap(self, ref_coords, query_coords, offsets, hint_key=None, symmetric=False):
        ref, qry, off = _cpu(ref_coords, "coords").int(), _cpu(query_coords, "coords").int(), _cpu(offsets, "offsets").int()
        hits = self.hash_query(self.kernel_hash(qry, off), self.hash(ref)).view(off.shape[0], -1)  # (K, Nq): ref row or -1
        pairs, sizes = [], []
        for k in range(off.shape[0]):
            q = torch.nonzero(hits[k] >= 0).squeeze(1)
            pairs.append(torch.stack([hits[k, q], q], dim=1))
            sizes.append(q.numel())
        return TorchKernelMap(torch.cat(pairs, dim=0) if pairs else torch.zeros((0, 2), dtype=torch.int64), sizes,
                              ref.shape[0], qry.shape[0])

    def tile_rows(self, cin, cout, kmap=None, dtype=0):
        return 128

    # -- convolution (TS:torchsparse/nn/functional/conv.py:67-79 semantics)
    def conv_gather_gemm(self, src, weight, kmap, bias=None, tile_rows=None, bn_sums=None, ordered=True):
        src, weight = _cpu(src, "input").float(), _cpu(weight, "weight").float()
        if src.shape[1] != weight.shape[1]:
            raise ValueError("Input feature size and kernel size mismatch")
        out = torch.zeros((kmap.n_dst, weight.shape[2]), dtype=torch.float32)
        ko, pr = kmap.koff_host, kmap.pairs.long()
        for k in range(kmap.K):
            if ko[k + 1] > ko[k]:
                p = pr[ko[k]:ko[k + 1]]
                out.index_add_(0, p[:, 1], src[p[:, 0]] @ weight[k])
        return out + bias if bias is not None else out

    def conv_wgrad(self, fa, fb, kmap, a_col, split=False):
        fa, fb = _cpu(fa, "input").float(), _cpu(fb, "grad_output").float()
        gw = torch.zeros((kmap.K, fa.shape[1], fb.shape[1]), dtype=torch.float32)
        ko, pr = kmap.koff_host, kmap.pairs.long()
        for k in range(kmap.K):
            if ko[k + 1] > ko[k]:
                p = pr[ko[k]:ko[k + 1]]
                gw[k] = fa[p[:, a_col]].t() @ fb[p[:, 1 - a_col]]
        return gw

    def transpose_weights(self, w):
        return _cpu(w, "weight").transpose(1, 2).contiguous()

    def conv_h_applies(self, cin, cout, k):
        return False

    def conv_x3_applies(self, cin, cout, k):
        return False

    # -- sparse_quantize (TS:torchsparse/utils/quantize.py:9-46): first occurrence per voxel, ascending ravel hash
    def quantize(self, points, voxel_size3, want_index, want_inverse):
        p = _cpu(points, "coords")
        vs = torch.tensor([float(v) for v in voxel_size3], dtype=torch.float64)
        c = torch.floor(p.double() / vs).int() if p.is_floating_point() else torch.floor(p.double() / vs).int()
        x = (c - c.min(dim=0).values).long()
        xmax = x.max(dim=0).values + 1
        h = (x[:, 0] * xmax[1] + x[:, 1]) * xmax[2] + x[:, 2]
        hs, order = torch.sort(h, stable=True)
        first = torch.ones_like(hs, dtype=torch.bool)
        first[1:] = hs[1:] != hs[:-1]
        index = order[first]
        inverse = torch.empty_like(h)
        inverse[order] = torch.cumsum(first.long(), 0) - 1
        return c[index].contiguous(), (index if want_index else None), (inverse if want_inverse else None)

    def unique_inverse_csr(self, keys):
        uniq, inverse, counts = torch.unique(_cpu(keys, "keys"), sorted=True, return_inverse=True, return_counts=True)
        return uniq, inverse, counts.int()

    # -- cylinder / range scatter (torch_scatter.scatter_max semantics as SURVEY.md section 8 a13 states them; RL:.../map_count.py, denselize.py)
    def scatter_max_fwd(self, src, index, m):
        src, index = _cpu(src, "src").float(), _cpu(index, "index").long()
        n, c = src.shape
        out = torch.full((m, c), float("-inf")).scatter_reduce(0, index.unsqueeze(1).expand(-1, c), src, "amax", include_self=True)
        rows = torch.arange(n).unsqueeze(1).expand(-1, c)
        cand = torch.where(src == out[index], rows, torch.full_like(rows, n))
        arg = torch.full((m, c), n, dtype=torch.int64).scatter_reduce(0, index.unsqueeze(1).expand(-1, c), cand, "amin", include_self=True)
        empty = arg == n
        out = torch.where(empty, torch.zeros_like(out), out)  # rows nobody scatters to: 0 / -1 (as the HIP kernel reports them)
        return out, torch.where(empty, torch.full_like(arg, -1), arg)

    def scatter_max_bwd(self, gout, arg, n):
        g, arg = _cpu(gout, "grad_output").float(), _cpu(arg, "arg").long()
        gs = torch.zeros((n + 1, g.shape[1]), dtype=torch.float32)
        gs.scatter_add_(0, torch.where(arg < 0, torch.full_like(arg, n), arg), g)
        return gs[:n].contiguous()

    def map_count(self, pxpy, b, h, w):
        p = _cpu(pxpy, "pxpy").long()
        ok = (p[:, 0] >= 0) & (p[:, 0] < b) & (p[:, 1] >= 0) & (p[:, 1] < w) & (p[:, 2] >= 0) & (p[:, 2] < h)
        flat = (p[ok, 0] * h + p[ok, 2]) * w + p[ok, 1]
        return torch.bincount(flat, minlength=b * h * w).view(b, h, w).int()

    def denselize_fwd(self, feat, count_map, pxpy):
        f, cm, p = _cpu(feat, "feat").float(), _cpu(count_map, "count_map"), _cpu(pxpy, "pxpy").long()
        (b, h, w), c = cm.shape, f.shape[1]
        ok = (p[:, 0] >= 0) & (p[:, 0] < b) & (p[:, 1] >= 0) & (p[:, 1] < w) & (p[:, 2] >= 0) & (p[:, 2] < h)
        flat = (p[ok, 0] * h + p[ok, 2]) * w + p[ok, 1]
        out = torch.zeros((b * h * w, c), dtype=torch.float32)
        out.index_add_(0, flat, f[ok] / cm.view(-1)[flat].float().unsqueeze(1))
        return out.view(b, h, w, c).permute(0, 3, 1, 2).contiguous()

    def denselize_bwd(self, gout, count_map, pxpy):
        g, cm, p = _cpu(gout, "top_grad").float(), _cpu(count_map, "count_map"), _cpu(pxpy, "pxpy").long()
        b, c, h, w = g.shape
        ok = (p[:, 0] >= 0) & (p[:, 0] < b) & (p[:, 1] >= 0) & (p[:, 1] < w) & (p[:, 2] >= 0) & (p[:, 2] < h)
        flat = (p[ok, 0] * h + p[ok, 2]) * w + p[ok, 1]
        rows = g.permute(0, 2, 3, 1).reshape(-1, c)
        gf = torch.zeros((p.shape[0], c), dtype=torch.float32)
        gf[ok] = rows[flat] / cm.view(-1)[flat].float().unsqueeze(1)
        return gf

    # -- BatchNorm pieces of the fused blocks (nn.BatchNorm1d training semantics), plain torch
    def bn_stats(self, x):
        xd = _cpu(x, "input").double()
        return torch.cat([xd.sum(0), (xd * xd).sum(0), torch.tensor([float(x.shape[0])], dtype=torch.float64)])

    def bn_finalize(self, sums, count, eps, momentum, running_mean, running_var, count_dev=None):
        c = sums.numel() // 2
        n = float(count_dev[0]) if count_dev is not None else float(count)
        mean = sums[:c] / n
        var = (sums[c:2 * c] / n - mean * mean).clamp_(min=0)
        if running_mean is not None:
            unbiased = var * n / (n - 1.0) if n > 1 else var
            running_mean.mul_(1 - momentum).add_((momentum * mean).float())
            running_var.mul_(1 - momentum).add_((momentum * unbiased).float())
        return torch.cat([mean, torch.rsqrt(var + eps)])

    @staticmethod
    def _mask_words(pos):
        n, c = pos.shape
        words = (pos.reshape(n, c // 32, 32).long() << torch.arange(32)).sum(-1)
        return torch.where(words >= 2 ** 31, words - 2 ** 32, words).int()

    @staticmethod
    def _gate(gate, c):
        if gate.dtype != torch.int32:
            return gate > 0
        return (((gate.long().unsqueeze(-1) >> torch.arange(32)) & 1) > 0).reshape(gate.shape[0], c)

    def bn_apply(self, x, res, stat, w, b, relu, want_mask=False, tail=None):
        c = x.shape[1]
        y = (_cpu(x, "input") - stat[:c].float()) * stat[c:].float()
        if w is not None:
            y = y * w + b
        if res is not None:
            y = y + res
        if relu:
            y = torch.relu(y)
        mask = self._mask_words(y > 0) if want_mask else None
        if tail is not None:
            y = torch.cat([y, tail], dim=1)
        return (y, mask) if want_mask else y

    def bn_bwd_stats(self, dy, x, gate, stat, relu):
        c = x.shape[1]
        g = dy * self._gate(gate, c) if relu else dy
        xh = (x - stat[:c].float()) * stat[c:].float()
        return torch.cat([g.double().sum(0), (g * xh).double().sum(0)])

    def bn_bwd_apply(self, dy, x, gate, stat, sums2, count, w, relu, want_res, count_dev=None):
        c = x.shape[1]
        n = float(count_dev[0]) if count_dev is not None else float(count)
        g = dy * self._gate(gate, c) if relu else dy
        xh = (x - stat[:c].float()) * stat[c:].float()
        dx = (g - (sums2[:c] / n).float() - xh * (sums2[c:] / n).float()) * stat[c:].float()
        if w is not None:
            dx = dx * w
        return dx, (g.clone() if want_res else None)


_SAVED = []


def install():
    """Make the pure-PyTorch CPU backend the process-wide backend (explicit opt-in; HIP tensors are refused by it)."""
    _SAVED.append(native._BACKEND)
    native._BACKEND = TorchCpuBackend()
    return native._BACKEND


def uninstall():
    native._BACKEND = _SAVED.pop() if _SAVED else None


@contextlib.contextmanager
def enabled():
    be = install()
    try:
        yield be
    finally:
        uninstall()


if os.environ.get("PCS_CPU_PATH") == "1" and native._BACKEND is None:
    install()
