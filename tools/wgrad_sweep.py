#!/usr/bin/env python3
"""fp32 weight-gradient (wgrad2) time per layer shape on the 12-frame bench maps, long timing (HIP events over many
launches after a warm-up). Usage: python tools/wgrad_sweep.py ["<level> <cin> <cout>" ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from openpcseg_amd import functional as F  # noqa: E402
from openpcseg_amd import native  # noqa: E402
from openpcseg_amd.workloads.synthetic import make_batch  # noqa: E402

DEFAULT = ["0 96 96", "1 96 96", "2 128 128", "2 64 64", "3 256 256", "3 128 128", "4 256 256"]


def main():
    reps = int(os.environ.get("PCS_SWEEP_REPS", "60"))
    dev = torch.device("cuda:0")
    coords = make_batch(list(range(12)))["lidar"].C.to(dev)
    coords = coords[torch.argsort(F.sphash(coords))].contiguous()
    levels, ts = [coords], 1
    for _ in range(4):
        levels.append(F.spdownsample(levels[-1], 2, 2, ts))
        ts *= 2
    be = native.backend()
    maps = {}
    warm = 200
    for spec in (sys.argv[1:] or DEFAULT):
        level, cin, cout = [int(v) for v in spec.split()]
        c = levels[level]
        if level not in maps:
            maps[level] = F.build_kernel_map(c, c, (3, 3, 3), (2 ** level,) * 3, (1, 1, 1))
        km = maps[level].fwd
        x = torch.randn(c.shape[0], cin, device=dev)
        gy = torch.randn(c.shape[0], cout, device=dev)
        for _ in range(warm):
            be.conv_wgrad(x, gy, km, 0)
        warm = 10
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            be.conv_wgrad(x, gy, km, 0)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        print("wgrad level=%d %dx%d: %.0f us  %.1f TFLOP/s" % (level, cin, cout, us, 2.0 * km.num_pairs * cin * cout / us / 1e6), flush=True)


if __name__ == "__main__":
    main()
