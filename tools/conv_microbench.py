#!/usr/bin/env python3
"""Time ONE fused-conv shape on real rulebooks of the synthetic scene.
Usage: python tools/conv_microbench.py <level 0..4> <cin> <cout> [reps] [frames] [tile_rows]
level 0 = stride 1 ... level 4 = stride 16 (k3 submanifold map at that level)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from openpcseg_amd import functional as F  # noqa: E402
from openpcseg_amd import native  # noqa: E402
from openpcseg_amd.workloads.synthetic import make_batch  # noqa: E402


def main():
    level, cin, cout = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
    frames = int(sys.argv[5]) if len(sys.argv) > 5 else 12
    tile = int(sys.argv[6]) if len(sys.argv) > 6 else None
    dev = torch.device("cuda:0")
    coords = make_batch(list(range(frames)))["lidar"].C.to(dev)
    h = F.sphash(coords)
    coords = coords[torch.argsort(h)].contiguous()  # stride-1 order of initial_voxelize
    ts = 1
    for _ in range(level):
        coords = F.spdownsample(coords, 2, 2, ts)
        ts *= 2
    entry = F.build_kernel_map(coords, coords, (3, 3, 3), (ts,) * 3, (1, 1, 1))
    n, p = coords.shape[0], entry.fwd.num_pairs
    be = native.backend()
    x = torch.randn(n, cin, device=dev)
    w = torch.randn(27, cin, cout, device=dev) * 0.05
    gy = torch.randn(n, cout, device=dev)
    for _ in range(2):
        be.conv_gather_gemm(x, w, entry.fwd, tile_rows=tile)
        be.conv_wgrad(x, gy, entry.fwd, 0)
    torch.cuda.synchronize()
    if tile is not None:  # cross-check the selected tile shape against the default path
        ref = be.conv_gather_gemm(x, w, entry.fwd, tile_rows=128)
        got = be.conv_gather_gemm(x, w, entry.fwd, tile_rows=tile)
        print("max rel diff vs tile 128: %.2e" % float((ref - got).abs().max() / ref.abs().max()))
    for name, fn in [("gemm", lambda: be.conv_gather_gemm(x, w, entry.fwd, tile_rows=tile)),
                     ("wgrad", lambda: be.conv_wgrad(x, gy, entry.fwd, 0))]:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        print("%s level=%d n=%d pairs=%d %d->%d tile=%s: %.0f us  %.1f TFLOP/s" %
              (name, level, n, p, cin, cout, tile, us, 2.0 * p * cin * cout / us / 1e6))


if __name__ == "__main__":
    main()
