#!/usr/bin/env python3
"""Two shapes of the fp32-on-bf16-MFMA fused convolution (conv_os5x_kernel), a few launches each, for counter collection /
ablation builds: stride 1 96->96 and stride 8 256->256 on the 12-frame bench maps. Prints HIP-event time per launch."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from openpcseg_amd import functional as F  # noqa: E402
from openpcseg_amd import native  # noqa: E402
from openpcseg_amd.workloads.synthetic import make_batch  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    warm = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    dev = torch.device("cuda:0")
    coords = make_batch(list(range(12)))["lidar"].C.to(dev)
    coords = coords[torch.argsort(F.sphash(coords))].contiguous()
    levels, ts = [coords], 1
    for _ in range(3):
        levels.append(F.spdownsample(levels[-1], 2, 2, ts))
        ts *= 2
    be = native.backend()
    for level, cin, cout in ((0, 96, 96), (2, 128, 128), (3, 256, 256)):
        c = levels[level]
        entry = F.build_kernel_map(c, c, (3, 3, 3), (2 ** level,) * 3, (1, 1, 1))
        n, p = c.shape[0], entry.fwd.num_pairs
        x = torch.randn(n, cin, device=dev)
        wp = be.prepare_weights_x3(torch.randn(27, cin, cout, device=dev) * 0.05, transpose=False)
        for _ in range(warm):
            be.conv_gather_gemm_x3(x, wp, 27, cout, entry.fwd)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            be.conv_gather_gemm_x3(x, wp, 27, cout, entry.fwd)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        print("convx level=%d n=%d pairs=%d %d->%d: %.0f us  %.1f TFLOP/s" % (level, n, p, cin, cout, us, 2.0 * p * cin * cout / us / 1e6), flush=True)


if __name__ == "__main__":
    main()
