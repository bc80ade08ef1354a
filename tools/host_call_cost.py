#!/usr/bin/env python3
"""Host-side cost (us per call) of the Python layer of the hot ops, on tiny tensors so that GPU time is negligible.
Usage: python tools/host_call_cost.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from openpcseg_amd import functional as F  # noqa: E402
from openpcseg_amd import native  # noqa: E402
from openpcseg_amd.fused import FusedBatchNorm  # noqa: E402
from openpcseg_amd.modules import Conv3d  # noqa: E402
from openpcseg_amd.sparse import SparseTensor  # noqa: E402


def cost(fn, n=300):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    dt = time.perf_counter() - t0
    torch.cuda.synchronize()
    return dt / n * 1e6


def main():
    dev = torch.device("cuda:0")
    be = native.backend()
    g = torch.Generator().manual_seed(0)
    coords = torch.unique(torch.cat([torch.randint(0, 12, (600, 3), generator=g), torch.zeros(600, 1, dtype=torch.long)], 1), dim=0).int().to(dev)
    n = coords.shape[0]
    x = torch.randn(n, 64, device=dev)
    w = torch.randn(27, 64, 64, device=dev)
    entry = F.build_kernel_map(coords, coords, (3, 3, 3), (1, 1, 1), (1, 1, 1))
    gy = torch.randn(n, 64, device=dev)
    rows = []
    rows.append(("native._stream()", cost(native._stream, 2000)))
    rows.append(("torch.empty((n,64))", cost(lambda: torch.empty((n, 64), device=dev), 2000)))
    rows.append(("be.hash", cost(lambda: be.hash(coords))))
    rows.append(("be.conv_gather_gemm", cost(lambda: be.conv_gather_gemm(x, w, entry.fwd))))
    rows.append(("be.conv_wgrad", cost(lambda: be.conv_wgrad(x, gy, entry.fwd, 0))))
    rows.append(("w.transpose(1,2).contiguous()", cost(lambda: w.transpose(1, 2).contiguous())))
    sums = be.bn_stats(x)
    stat = be.bn_finalize(sums, float(n), 1e-5, 0.1, None, None)
    rows.append(("be.bn_stats", cost(lambda: be.bn_stats(x))))
    rows.append(("be.bn_finalize", cost(lambda: be.bn_finalize(sums, float(n), 1e-5, 0.1, None, None))))
    rows.append(("be.bn_apply (+mask)", cost(lambda: be.bn_apply(x, None, stat, None, None, True, want_mask=True))))
    conv = Conv3d(64, 64, 3).to(dev)
    bn = FusedBatchNorm(64).to(dev).train()
    st = SparseTensor(x, coords)
    conv(st)  # builds + caches the map
    rows.append(("Conv3d forward (cached map, no grad)", cost(lambda: torch.no_grad().__enter__() or conv(st))))
    torch.set_grad_enabled(True)
    rows.append(("FusedBatchNorm forward (no grad)", cost(lambda: torch.no_grad().__enter__() or bn(st, relu=True))))
    torch.set_grad_enabled(True)
    xg = x.clone().requires_grad_(True)

    def fb():
        s2 = SparseTensor(xg, coords)
        s2.cmaps, s2.kmaps = st.cmaps, st.kmaps
        y = bn(conv(s2), relu=True)
        y.F.sum().backward()
    rows.append(("Conv3d + FusedBN forward+backward (autograd)", cost(fb, 200)))
    for name, us in rows:
        print("%-48s %8.1f us" % (name, us))


if __name__ == "__main__":
    main()
