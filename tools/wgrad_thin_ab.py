"""A/B of the weight gradient of the thin layers under mixed precision (bf16 operands): wgrad2<Bf16> (fp32 MFMA, PCS_WGRAD3_THIN=0)
vs the one-wave wgrad3 instance (16-bit MFMA). Run twice:  PCS_WGRAD3_THIN=0 python tools/wgrad_thin_ab.py ; python tools/wgrad_thin_ab.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
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
    for s in (1, 2, 4):
        lv[2 * s] = be.downsample(lv[s], [2 * s] * 3)
    print("PCS_WGRAD3_THIN=%s" % os.environ.get("PCS_WGRAD3_THIN", "1"))
    tot = 0.0
    for s, cin, cout, n_layers in [(1, 32, 32, 1), (2, 32, 32, 4), (4, 32, 64, 1), (4, 64, 64, 5)]:
        c = lv[s]
        entry = F.build_kernel_map(c, c, (3, 3, 3), (s,) * 3, (1, 1, 1))
        p = entry.fwd.num_pairs
        xh = torch.randn(c.shape[0], cin, device=dev).bfloat16()
        gh = torch.randn(c.shape[0], cout, device=dev).bfloat16()
        t = timed(lambda: be.conv_wgrad_h(xh, gh, entry.fwd, 0))
        tot += t * n_layers
        print("stride %d  %3d x %3d  P %8d : %.3f ms  %.1f TFLOP/s  (x %d layers)" % (s, cin, cout, p, t, 2.0 * p * cin * cout / t / 1e9, n_layers))
    print("per step: %.2f ms" % tot)


if __name__ == "__main__":
    main()
