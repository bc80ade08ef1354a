"""The reference's native module surface `torchsparse.backend`: all 20 names of
TS:torchsparse/backend/pybind_cuda.cpp:18-39 -- the `*_cuda` half on top of the C ABI, the `*_cpu` half
(TS:torchsparse/backend/pybind_cpu.cpp:12-23) on this package's pure-PyTorch CPU path (cpu_fallback.py, BASELINE
config 1) -- so that code written against the reference's low-level functions (its own nn/functional/*.py, run
unmodified by tests/test_backend_shim.py on both device types) keeps working. Same argument order, ownership and
result conventions as the reference (boundary B-B).

This is a COMPATIBILITY surface, not the fast path: the reference's (nbmaps, nbsizes-on-the-host) calling convention
costs a host read, a sort and three small host-to-device copies per convolution call (`_as_kmap`). The fast path is
openpcseg_amd.functional / `install_as_torchsparse()`, which keeps the native maps."""
import torch

from . import native
from .native import KernelMap


def hash_cuda(idx):
    return native.backend().hash(idx)


def kernel_hash_cuda(idx, kernel_offset):
    return native.backend().kernel_hash(idx, kernel_offset)


def hash_query_cuda(hash_query, hash_target, idx_target):
    """-> idx_target[position] + 1, or 0 when the query hash is absent (query_cuda.cu:9-56)."""
    be = native.backend()
    pos = be.table_query(be.table_build(hash_target), hash_query)  # position + 1 / 0
    hit = pos > 0
    vals = idx_target[(pos - 1).clamp_(min=0)] + 1
    return torch.where(hit, vals, torch.zeros_like(vals))


def count_cuda(idx, s):
    return native.backend().count(idx, s)


def voxelize_forward_cuda(inputs, idx, counts):
    return native.backend().voxelize_fwd(inputs, idx, counts)


def voxelize_backward_cuda(top_grad, idx, counts, n):
    return native.backend().voxelize_bwd(top_grad, idx, counts, n)


def devoxelize_forward_cuda(feat, indices, weight):
    return native.backend().devoxelize_fwd(feat, indices, weight)


def devoxelize_backward_cuda(top_grad, indices, weight, n):
    return native.backend().devoxelize_bwd(top_grad, indices, weight, n)


def _as_kmap(neighbor_map, neighbor_offset, n_src, n_dst, dst_col):
    """Wrap the reference's (nbmaps (P,2) int32 [in,out], nbsizes (K,) int32 ON CPU) as the native map with the
    destination rows in column 1, sorted ascending inside every offset (a stable per-offset sort when the
    caller's column is not already sorted, i.e. for transposed use)."""
    sizes = [int(v) for v in neighbor_offset.tolist()]
    koff_host = [0]
    for s in sizes:
        koff_host.append(koff_host[-1] + s)
    pairs = neighbor_map.int()
    if dst_col == 0:
        pairs = pairs[:, [1, 0]]
        kid = torch.repeat_interleave(torch.arange(len(sizes), device=pairs.device),
                                      torch.tensor(sizes, device=pairs.device))
        key = kid * (int(n_dst) + 1) + pairs[:, 1].long()
        pairs = pairs[torch.argsort(key, stable=True)]
    pairs = pairs.contiguous()
    dev = pairs.device
    return KernelMap(pairs, torch.tensor(koff_host, dtype=torch.int32, device=dev), koff_host,
                     torch.tensor(sizes, dtype=torch.int64, device=dev), n_src, n_dst)


def convolution_forward_cuda(in_feat, out_feat, kernel, neighbor_map, neighbor_offset, transpose):
    """out_feat (pre-allocated by the caller, convolution_cuda.cu:53-165) is overwritten in place."""
    if in_feat.size(1) != kernel.size(1):
        raise ValueError("Input feature size and kernel size mismatch")
    km = _as_kmap(neighbor_map, neighbor_offset, in_feat.shape[0], out_feat.shape[0], 0 if transpose else 1)
    out_feat.copy_(native.backend().conv_gather_gemm(in_feat.contiguous(), kernel.contiguous(), km))


def convolution_backward_cuda(in_feat, grad_in_feat, grad_out_feat, kernel, grad_kernel, neighbor_map,
                              neighbor_offset, transpose):
    """grad_in_feat / grad_kernel are resized + overwritten like convolution_cuda.cu:167-278."""
    be = native.backend()
    # dgrad: destination = rows of in_feat (map column `transpose`), source = rows of grad_out
    km = _as_kmap(neighbor_map, neighbor_offset, grad_out_feat.shape[0], in_feat.shape[0], 1 if transpose else 0)
    grad_in_feat.resize_as_(in_feat).copy_(
        be.conv_gather_gemm(grad_out_feat.contiguous(), kernel.transpose(1, 2).contiguous(), km))
    kw = _as_kmap(neighbor_map, neighbor_offset, in_feat.shape[0], grad_out_feat.shape[0], 1)
    grad_kernel.resize_as_(kernel).copy_(be.conv_wgrad(in_feat.contiguous(), grad_out_feat.contiguous(), kw,
                                                       1 if transpose else 0))


# ---- the `*_cpu` half (TS:torchsparse/backend/pybind_cpu.cpp:12-23): host tensors, pure PyTorch -------------------------------
def _cpu_be():
    from .cpu_fallback import TorchCpuBackend
    return TorchCpuBackend()


def hash_cpu(idx):
    return _cpu_be().hash(idx)


def kernel_hash_cpu(idx, kernel_offset):
    return _cpu_be().kernel_hash(idx, kernel_offset)


def hash_query_cpu(hash_query, hash_target, idx_target):
    """-> idx_target[position] + 1, or 0 when the query hash is absent (TS:torchsparse/backend/others/query_cpu.cpp: the first of
    equal targets wins)."""
    pos = _cpu_be().hash_query(hash_query, hash_target)
    if idx_target.numel() == 0:
        return torch.zeros_like(pos)
    vals = idx_target[pos.clamp(min=0)] + 1
    return torch.where(pos >= 0, vals, torch.zeros_like(vals))


def count_cpu(idx, s):
    return _cpu_be().count(idx, s)


def voxelize_forward_cpu(inputs, idx, counts):
    return _cpu_be().voxelize_fwd(inputs, idx, counts)


def voxelize_backward_cpu(top_grad, idx, counts, n):
    return _cpu_be().voxelize_bwd(top_grad, idx, counts, n)


def devoxelize_forward_cpu(feat, indices, weight):
    return _cpu_be().devoxelize_fwd(feat, indices, weight)


def devoxelize_backward_cpu(top_grad, indices, weight, n):
    """The gradient of devoxelize_forward (the reference's own CPU twin is broken here -- SURVEY.md section 8c: it indexes
    top_grad by the voxel index; authority = TS:torchsparse/backend/devoxelize/devoxelize_cuda.cu:37-57)."""
    return _cpu_be().devoxelize_bwd(top_grad, indices, weight, n)


def _offset_slices(neighbor_offset):
    a = 0
    for k, n in enumerate(int(v) for v in neighbor_offset.tolist()):
        yield k, a, a + n
        a += n


def convolution_forward_cpu(in_feat, out_feat, kernel, neighbor_map, neighbor_offset, transpose):
    """out_feat (zeros from the caller) accumulates per offset `out[o] += in[i] @ W[k]` (convolution_cpu.cpp:38-91 =
    TS:torchsparse/nn/functional/conv.py:67-79)."""
    if in_feat.size(1) != kernel.size(1):
        raise ValueError("Input feature size and kernel size mismatch")
    nm = neighbor_map.long()
    ci, co = (1, 0) if transpose else (0, 1)
    for k, a, b in _offset_slices(neighbor_offset):
        if b > a:
            out_feat.index_add_(0, nm[a:b, co], in_feat[nm[a:b, ci]] @ kernel[k])


def convolution_backward_cpu(in_feat, grad_in_feat, grad_out_feat, kernel, grad_kernel, neighbor_map, neighbor_offset, transpose):
    """grad_in_feat / grad_kernel are resized + overwritten like convolution_cpu.cpp:93-183."""
    nm = neighbor_map.long()
    ci, co = (1, 0) if transpose else (0, 1)
    grad_in_feat.resize_as_(in_feat).zero_()
    grad_kernel.resize_as_(kernel).zero_()
    for k, a, b in _offset_slices(neighbor_offset):
        if b > a:
            i, o = nm[a:b, ci], nm[a:b, co]
            g = grad_out_feat[o]
            grad_in_feat.index_add_(0, i, g @ kernel[k].t())
            grad_kernel[k] = in_feat[i].t() @ g
