"""Containers and small helpers of the reference operator API.

Mirrors (same names, attributes and cache-sharing semantics):
  SparseTensor / PointTensor   TS:torchsparse/tensor.py:10-105
  cat                          TS:torchsparse/operators.py:10-17
  fapply                       TS:torchsparse/nn/utils/apply.py:10-16
  get_kernel_offsets           TS:torchsparse/nn/utils/kernel.py:11-32
  make_ntuple                  TS:torchsparse/utils/utils.py:9-19
The `cmaps` / `kmaps` dicts are shared BY REFERENCE between a tensor and everything derived
from it -- that sharing is the rulebook cache of one forward pass.
"""
from itertools import repeat

import numpy as np
import torch

__all__ = ["SparseTensor", "PointTensor", "cat", "fapply", "get_kernel_offsets", "make_ntuple"]


def make_ntuple(x, ndim):
    if isinstance(x, int):
        x = tuple(repeat(x, ndim))
    elif isinstance(x, list):
        x = tuple(x)
    elif isinstance(x, torch.Tensor):
        x = tuple(int(v) for v in x.reshape(-1).cpu().tolist())
    assert isinstance(x, tuple) and len(x) == ndim, x
    return x


class SparseTensor:
    """feats (N,C) + coords (N,4) int32 [x,y,z,batch] + 3-tuple stride + shared map caches."""

    def __init__(self, feats, coords, stride=1):
        self.feats = feats
        self.coords = coords
        self.stride = make_ntuple(stride, ndim=3)
        self.cmaps = {}
        self.kmaps = {}

    # short aliases used all over the segmentors
    F = property(lambda self: self.feats, lambda self, v: setattr(self, "feats", v))
    C = property(lambda self: self.coords, lambda self, v: setattr(self, "coords", v))
    s = property(lambda self: self.stride,
                 lambda self, v: setattr(self, "stride", make_ntuple(v, ndim=3)))

    def _map(self, fn):
        self.coords = fn(self.coords)
        self.feats = fn(self.feats)
        return self

    def cpu(self):
        return self._map(lambda t: t.cpu())

    def cuda(self):
        return self._map(lambda t: t.cuda())

    def detach(self):
        return self._map(lambda t: t.detach())

    def to(self, device, non_blocking=True):
        return self._map(lambda t: t.to(device, non_blocking=non_blocking))

    def _like(self, feats):
        out = SparseTensor(feats, self.coords, self.stride)
        out.cmaps = self.cmaps
        out.kmaps = self.kmaps
        return out

    def __add__(self, other):
        return self._like(self.feats + other.feats)


class PointTensor:
    """Per-point features + float coords, with the voxel<->point caches of one forward."""

    def __init__(self, feats, coords, idx_query=None, weights=None):
        self.F = feats
        self.C = coords
        self.idx_query = idx_query if idx_query is not None else {}
        self.weights = weights if weights is not None else {}
        self.additional_features = {"idx_query": {}, "counts": {}}

    def _map(self, fn):
        self.F = fn(self.F)
        self.C = fn(self.C)
        return self

    def cuda(self):
        return self._map(lambda t: t.cuda())

    def detach(self):
        return self._map(lambda t: t.detach())

    def to(self, device, non_blocking=True):
        return self._map(lambda t: t.to(device, non_blocking=non_blocking))

    def __add__(self, other):
        out = PointTensor(self.F + other.F, self.C, self.idx_query, self.weights)
        out.additional_features = self.additional_features
        return out


def cat(inputs):
    """Channel-concatenate SparseTensors that share coordinates."""
    fused = getattr(inputs[0], "cat_with", None)   # block_fusion.PendingBatchNorm: the BatchNorm apply pass writes the concatenation
    if fused is not None and len(inputs) == 2:
        out = fused(inputs[1])
        if out is not None:
            return out
    return inputs[0]._like(torch.cat([x.feats for x in inputs], dim=1))


def fapply(input, fn, *args, **kwargs):
    """Apply a dense function to the features, keeping coords / stride / caches."""
    return input._like(fn(input.feats, *args, **kwargs))


def get_kernel_offsets(size, stride=1, dilation=1, device="cpu"):
    """(K,3) int32 kernel offsets. Odd volume: x fastest, then y, then z; even volume: z
    fastest (kernel.py:24-29) -- this order IS the weight-slice <-> offset mapping."""
    size = make_ntuple(size, ndim=3)
    stride = make_ntuple(stride, ndim=3)
    dilation = make_ntuple(dilation, ndim=3)
    axes = [np.arange(-size[d] // 2 + 1, size[d] // 2 + 1) * stride[d] * dilation[d] for d in range(3)]
    if int(np.prod(size)) % 2 == 1:
        grid = [[x, y, z] for z in axes[2] for y in axes[1] for x in axes[0]]
    else:
        grid = [[x, y, z] for x in axes[0] for y in axes[1] for z in axes[2]]
    if torch.device(device).type == "cpu":
        return torch.tensor(grid, dtype=torch.int)
    # pinned staging + asynchronous copy: a pageable host-to-device copy would wait for the whole GPU queue
    return torch.tensor(grid, dtype=torch.int, pin_memory=True).to(device, non_blocking=True)
