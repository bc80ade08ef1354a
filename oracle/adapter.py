"""Torch-facing adapters that give the oracle (and the reference's compiled CPU backend) the
same method surface as openpcseg_amd.native.HipBackend, on CPU tensors.

TEST INFRASTRUCTURE ONLY. tests/ monkeypatch `openpcseg_amd.native._BACKEND` with one of these
to exercise the host logic (containers, conv3d dispatcher, autograd wiring, DDP sharding) on a
machine without a GPU, and to produce the expected values the HIP path is compared with;
bench.py's cpu_baseline leg uses RefBackend to time the reference's own CPU code.
The product never imports this module.
"""
import numpy as np
import torch

from . import build_ref, oracle as orc


def _np(t):
    return t.detach().cpu().contiguous().numpy()


class CpuKernelMap:
    def __init__(self, pairs, nbsizes, n_src, n_dst):
        self.pairs = torch.from_numpy(np.ascontiguousarray(pairs, dtype=np.int32))
        self.nbsizes = torch.from_numpy(np.ascontiguousarray(nbsizes, dtype=np.int64))
        sizes = [int(s) for s in nbsizes]
        self.koff_host = [0]
        for s in sizes:
            self.koff_host.append(self.koff_host[-1] + s)
        self.koff = torch.tensor(self.koff_host, dtype=torch.int32)
        self.n_src, self.n_dst, self.K = n_src, n_dst, len(sizes)

    @property
    def num_pairs(self):
        return self.koff_host[-1]

    def mirror(self):  # same contract as native.KernelMap.mirror
        k, ko = self.K, self.koff_host
        pairs = np.concatenate([self.pairs.numpy()[ko[k - 1 - j]:ko[k - j]] for j in range(k)], axis=0).reshape(-1, 2)
        return CpuKernelMap(pairs, self.nbsizes.numpy()[::-1].copy(), self.n_dst, self.n_src)


class OracleBackend:
    """CPU restatement (oracle/pcs_oracle.c + oracle/oracle.py)."""

    name = "oracle-cpu"

    def hash(self, coords):
        return torch.from_numpy(orc.sphash(_np(coords)))

    def kernel_hash(self, coords, offsets):
        return torch.from_numpy(orc.sphash(_np(coords), _np(offsets)))

    def hash_query(self, queries, references):
        return torch.from_numpy(orc.sphashquery(_np(queries), _np(references)))

    def count(self, idx, num):
        return torch.from_numpy(orc.spcount(_np(idx), num))

    def voxelize_fwd(self, feats, idx, counts, cache_on=None):
        return torch.from_numpy(orc.voxelize_fwd(_np(feats), _np(idx), _np(counts)))

    def voxelize_bwd(self, gout, idx, counts, n):
        return torch.from_numpy(orc.voxelize_bwd(_np(gout), _np(idx), _np(counts), n))

    def devoxelize_fwd(self, feats, idx8, w8):
        return torch.from_numpy(orc.devoxelize_fwd(_np(feats), _np(idx8), _np(w8)))

    def devoxelize_bwd(self, gout, idx8, w8, m):
        return torch.from_numpy(orc.devoxelize_bwd(_np(gout), _np(idx8), _np(w8), m))

    def ti_weights(self, coords, idx_query, scale):
        return torch.from_numpy(orc.calc_ti_weights(_np(coords), _np(idx_query), scale))

    def transpose_weights(self, w):
        return w.transpose(1, 2).contiguous()

    def quantize(self, points, voxel_size3, want_index, want_inverse):
        vox, index, inverse = orc.sparse_quantize(_np(points), voxel_size3)
        return (torch.from_numpy(vox), torch.from_numpy(index) if want_index else None,
                torch.from_numpy(inverse) if want_inverse else None)

    def downsample(self, coords, sample_stride, offsets=None, coords_min=None):
        """TS:torchsparse/nn/functional/downsample.py:25-51 given the already-derived
        sample_stride / offsets / coords_min (what the product's backend receives)."""
        c = _np(coords).astype(np.int32)
        ss = np.asarray(sample_stride, dtype=np.int32)[None, :]
        if offsets is None:
            out = c.copy()
            q = np.trunc(c[:, :3].astype(np.float32) / ss.astype(np.float32))
            out[:, :3] = (q * ss.astype(np.float32)).astype(np.int32)
        else:
            off = _np(offsets).astype(np.int32)
            cmin = _np(coords_min).astype(np.int32)[None, :]
            x = (c[:, None, :3] + off[None, :, :]).reshape(-1, 3)
            b = np.repeat(c[:, 3:], off.shape[0], axis=1).reshape(-1, 1)
            cand = np.concatenate([x, b], axis=1)
            mask = (np.mod(cand[:, :3], ss) == 0) & (cand[:, :3] >= cmin)
            out = cand[mask.all(axis=1)]
        out = np.unique(out[:, [3, 0, 1, 2]], axis=0)
        return torch.from_numpy(np.ascontiguousarray(out[:, [1, 2, 3, 0]], dtype=np.int32))

    def build_kmap(self, ref_coords, query_coords, offsets, hint_key=None, symmetric=False):
        nbmaps, nbsizes = orc.build_kmap(_np(ref_coords), _np(query_coords), None, offsets=_np(offsets))
        return CpuKernelMap(nbmaps, nbsizes, ref_coords.shape[0], query_coords.shape[0])

    def tile_rows(self, cin, cout):
        return 128

    def conv_gather_gemm(self, src, weight, kmap, bias=None, tile_rows=None, bn_sums=None):
        if src.shape[1] != weight.shape[1]:
            raise ValueError("Input feature size and kernel size mismatch")
        out = orc.conv_fwd(_np(src), _np(weight), _np(kmap.pairs), _np(kmap.nbsizes),
                           (kmap.n_src, kmap.n_dst), transposed=False)
        out = torch.from_numpy(out)
        return out + bias if bias is not None else out

    def conv_wgrad(self, fa, fb, kmap, a_col, split=False):
        k, ca, cb = kmap.K, fa.shape[1], fb.shape[1]
        w0 = np.zeros((k, ca, cb), dtype=np.float32)
        _, gw = orc.conv_bwd(_np(fa), _np(fb), w0, _np(kmap.pairs), _np(kmap.nbsizes), transposed=bool(a_col))
        return torch.from_numpy(gw)


    def scatter_max_fwd(self, src, index, m):
        out, arg = orc.scatter_max(_np(src), _np(index), m)
        return torch.from_numpy(out), torch.from_numpy(arg)

    def scatter_max_bwd(self, gout, arg, n):
        return torch.from_numpy(orc.scatter_max_bwd(_np(gout), _np(arg), n))

    def map_count(self, pxpy, b, h, w):
        return torch.from_numpy(orc.map_count(_np(pxpy), b, h, w))

    def denselize_fwd(self, feat, count_map, pxpy):
        return torch.from_numpy(orc.denselize_fwd(_np(feat), _np(count_map), _np(pxpy)))

    def denselize_bwd(self, gout, count_map, pxpy):
        return torch.from_numpy(orc.denselize_bwd(_np(gout), _np(count_map), _np(pxpy)))


    # fused BatchNorm pieces restated with plain torch (nn.BatchNorm1d training semantics)
    def bn_stats(self, x):
        xd = x.double()
        return torch.cat([xd.sum(0), (xd * xd).sum(0), torch.tensor([float(x.shape[0])], dtype=torch.float64)])

    def bn_finalize(self, sums, count, eps, momentum, running_mean, running_var, count_dev=None):
        c = sums.numel() // 2
        if count_dev is not None:
            count = float(count_dev[0])
        mean = sums[:c] / count
        var = (sums[c:2 * c] / count - mean * mean).clamp_(min=0)
        if running_mean is not None:
            unb = var * count / (count - 1.0) if count > 1 else var
            running_mean.mul_(1 - momentum).add_((momentum * mean).float())
            running_var.mul_(1 - momentum).add_((momentum * unb).float())
        return torch.cat([mean, 1.0 / torch.sqrt(var + eps)])

    def bn_apply(self, x, res, stat, w, b, relu, want_mask=False, tail=None):
        if tail is not None:
            got = self.bn_apply(x, res, stat, w, b, relu, want_mask)
            y = got[0] if want_mask else got
            y = torch.cat([y, tail], dim=1)
            return (y, got[1]) if want_mask else y
        c = x.shape[1]
        y = (x - stat[:c].float()) * stat[c:].float()
        if w is not None:
            y = y * w + b
        if res is not None:
            y = y + res
        y = torch.relu(y) if relu else y
        if not want_mask:
            return y
        bits = (y > 0).reshape(y.shape[0], c // 32, 32).long() << torch.arange(32)
        words = bits.sum(-1)
        return y, torch.where(words >= 2 ** 31, words - 2 ** 32, words).int()  # bit ch % 32 of word ch / 32

    @staticmethod
    def _gate(gate, c):
        if gate.dtype != torch.int32:
            return gate > 0
        return (((gate.long().unsqueeze(-1) >> torch.arange(32)) & 1) > 0).reshape(gate.shape[0], c)

    def bn_bwd_stats(self, dy, x, gate, stat, relu):
        c = x.shape[1]
        g = (dy * self._gate(gate, c)) if relu else dy
        xh = (x - stat[:c].float()) * stat[c:].float()
        return torch.cat([g.double().sum(0), (g * xh).double().sum(0)])

    def bn_bwd_apply(self, dy, x, gate, stat, sums2, count, w, relu, want_res, count_dev=None):
        c = x.shape[1]
        if count_dev is not None:
            count = float(count_dev[0])
        g = (dy * self._gate(gate, c)) if relu else dy
        xh = (x - stat[:c].float()) * stat[c:].float()
        dx = (g - (sums2[:c] / count).float() - xh * (sums2[c:] / count).float()) * stat[c:].float()
        if w is not None:
            dx = dx * w
        return dx, (g.clone() if want_res else None)


class RefBackend(OracleBackend):
    """The reference's OWN compiled CPU functions (oracle/_ref) wherever its twin is sound
    (SURVEY.md section 8c); the restatement for kernel_hash (multi-batch bug, hash_cpu.cpp:29)
    and devoxelize backward (devoxelize_cpu.cpp:51-53)."""

    name = "reference-cpu"

    def __init__(self):
        self.ref = build_ref.load()
        if self.ref is None:
            raise RuntimeError("oracle/_ref is not built (run oracle/build_ref.py where /root/reference exists)")

    def hash(self, coords):
        return self.ref.hash_cpu(coords.contiguous())

    def hash_query(self, queries, references):
        idx = torch.arange(references.numel(), dtype=torch.long)
        return self.ref.hash_query_cpu(queries.reshape(-1).contiguous(), references.contiguous(), idx) - 1

    def count(self, idx, num):
        return self.ref.count_cpu(idx.contiguous(), int(num))

    def voxelize_fwd(self, feats, idx, counts, cache_on=None):
        return self.ref.voxelize_forward_cpu(feats.contiguous(), idx.contiguous(), counts.contiguous())

    def voxelize_bwd(self, gout, idx, counts, n):
        return self.ref.voxelize_backward_cpu(gout.contiguous(), idx.contiguous(), counts.contiguous(), n)

    def devoxelize_fwd(self, feats, idx8, w8):
        return self.ref.devoxelize_forward_cpu(feats.contiguous(), idx8.contiguous(), w8.contiguous())

    def conv_gather_gemm(self, src, weight, kmap, bias=None, tile_rows=None, bn_sums=None):
        out = torch.zeros(kmap.n_dst, weight.shape[-1])
        self.ref.convolution_forward_cpu(src.contiguous(), out, weight.contiguous(), kmap.pairs,
                                         kmap.nbsizes.int(), False)
        return out + bias if bias is not None else out

    def conv_wgrad(self, fa, fb, kmap, a_col, split=False):
        k, ca, cb = kmap.K, fa.shape[1], fb.shape[1]
        gin = torch.zeros_like(fa)
        gw = torch.zeros(k, ca, cb)
        self.ref.convolution_backward_cpu(fa.contiguous(), gin, fb.contiguous(), torch.zeros(k, ca, cb), gw,
                                          kmap.pairs, kmap.nbsizes.int(), bool(a_col))
        return gw
