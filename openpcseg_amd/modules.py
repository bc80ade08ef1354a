"""nn.Module layer of the reference operator API (`torchsparse.nn`).

  Conv3d      TS:torchsparse/nn/modules/conv.py:15-72   (parameter names `kernel` / `bias`,
              kernel shape (K,Cin,Cout) or (Cin,Cout) for K == 1, init U(+-1/sqrt(fan*K)) -- the
              checkpoint layout contract)
  BatchNorm   TS:torchsparse/nn/modules/norm.py:10-13
  ReLU / LeakyReLU   TS:torchsparse/nn/modules/activation.py:9-18
"""
import math

import numpy as np
import torch
from torch import nn

from . import functional as F
from .sparse import fapply, make_ntuple

__all__ = ["Conv3d", "BatchNorm", "ReLU", "LeakyReLU"]


class Conv3d(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, dilation=1, bias=False,
                 transposed=False):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.kernel_size = make_ntuple(kernel_size, ndim=3)
        self.stride = make_ntuple(stride, ndim=3)
        self.dilation = dilation
        self.transposed = transposed
        self.kernel_volume = int(np.prod(self.kernel_size))
        shape = (in_channels, out_channels)
        if self.kernel_volume > 1:
            shape = (self.kernel_volume,) + shape
        self.kernel = nn.Parameter(torch.zeros(*shape))
        if bias:
            self.bias = nn.Parameter(torch.zeros(out_channels))
        else:
            self.register_parameter("bias", None)
        # fused blocks (openpcseg_amd.fused) set this on convolutions that feed a FusedBatchNorm: the convolution then
        # also hands over the BatchNorm statistics of its output (computed in its write-back)
        self.emit_bn_stats = False
        self.reset_parameters()

    def reset_parameters(self):
        fan = self.out_channels if self.transposed else self.in_channels
        std = 1.0 / math.sqrt(fan * self.kernel_volume)
        self.kernel.data.uniform_(-std, std)
        if self.bias is not None:
            self.bias.data.uniform_(-std, std)

    def extra_repr(self):
        parts = ["%d -> %d" % (self.in_channels, self.out_channels), "k=%s" % (tuple(self.kernel_size),)]
        if any(v != 1 for v in self.stride):
            parts.append("stride=%s" % (tuple(self.stride),))
        if self.dilation != 1:
            parts.append("dilation=%s" % (self.dilation,))
        parts.append("bias" if self.bias is not None else "no bias")
        if self.transposed:
            parts.append("transposed")
        return ", ".join(parts)

    def forward(self, input, with_skip=False):
        """with_skip (not in the reference's signature; used by the fused residual blocks): -> (output, input routed through this
        convolution's autograd node), see functional.conv3d."""
        kw = {"with_skip": True} if with_skip else {}
        if self.emit_bn_stats and self.training:
            return F.conv3d(input, self.kernel, kernel_size=self.kernel_size, bias=self.bias, stride=self.stride,
                            dilation=self.dilation, transposed=self.transposed, bn_stats=True, **kw)
        return F.conv3d(input, self.kernel, kernel_size=self.kernel_size, bias=self.bias,
                        stride=self.stride, dilation=self.dilation, transposed=self.transposed, **kw)


class BatchNorm(nn.BatchNorm1d):
    def forward(self, input):
        return fapply(input, super().forward)


class ReLU(nn.ReLU):
    def forward(self, input):
        return fapply(input, super().forward)


class LeakyReLU(nn.LeakyReLU):
    def forward(self, input):
        return fapply(input, super().forward)
