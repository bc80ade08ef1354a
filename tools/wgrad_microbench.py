"""Weight-gradient time per layer shape on the 12-frame bench maps (HIP events, median of 5), fp32 and bf16 operands.
PCS_WGRAD3=0 python tools/wgrad_microbench.py   (fp32-MFMA kernel wgrad2)   vs   python tools/wgrad_microbench.py (wgrad3)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openpcseg_amd import functional as F  # noqa: E402
from openpcseg_amd import native  # noqa: E402
from openpcseg_amd.workloads.synthetic import make_batch  # noqa: E402
from half_conv_microbench import timed  # noqa: E402


def main():
    be = native.backend()
    dev = "cuda"
    c1 = make_batch(list(range(12)))["lidar"].C.to(dev)
    c1 = c1[torch.argsort(be.hash(c1))].contiguous()
    lv = {1: c1}
    for s in (1, 2, 4, 8):
        lv[2 * s] = be.downsample(lv[s], [2 * s] * 3)
    shapes = [(1, 32, 32), (1, 96, 96), (1, 128, 96), (2, 64, 64), (2, 96, 96), (4, 128, 128), (4, 192, 128), (8, 256, 256),
              (8, 384, 256), (16, 256, 256)]
    print("wgrad kernel: %s" % ("wgrad2 (fp32 MFMA)" if os.environ.get("PCS_WGRAD3") == "0" else "wgrad3 (16-bit MFMA)"))
    print("| stride | N | P | cin x cout | fp32 ms | fp32 TFLOP/s | fp32 via bf16x3 ms | bf16 ms | bf16 TFLOP/s |")
    print("|---|---|---|---|---|---|---|---|---|")
    tot32 = tot16 = 0.0
    for s, cin, cout in shapes:
        c = lv[s]
        entry = F.build_kernel_map(c, c, (3, 3, 3), (s,) * 3, (1, 1, 1))
        n, p = c.shape[0], entry.fwd.num_pairs
        x = torch.randn(n, cin, device=dev)
        gy = torch.randn(n, cout, device=dev)
        t32 = timed(lambda: be.conv_wgrad(x, gy, entry.fwd, 0))
        t3 = timed(lambda: be.conv_wgrad(x, gy, entry.fwd, 0, split=True))
        xh, gh = x.bfloat16(), gy.bfloat16()
        t16 = timed(lambda: be.conv_wgrad_h(xh, gh, entry.fwd, 0))
        fl = 2.0 * p * cin * cout
        tot32 += t32
        tot16 += t16
        print("| %d | %d | %d | %d x %d | %.3f | %.1f | %.3f | %.3f | %.1f |" % (s, n, p, cin, cout, t32, fl / t32 / 1e9, t3, t16, fl / t16 / 1e9))
    print("total: fp32 %.2f ms, bf16 %.2f ms" % (tot32, tot16))


if __name__ == "__main__":
    main()
