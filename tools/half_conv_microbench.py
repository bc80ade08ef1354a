"""fp32 vs bf16 / fp16 fused-conv time per layer shape on the 12-frame bench maps (HIP events, median of 5):
python tools/half_conv_microbench.py  ->  markdown table on stdout."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openpcseg_amd import functional as F  # noqa: E402
from openpcseg_amd import native  # noqa: E402
from openpcseg_amd.workloads.synthetic import make_batch  # noqa: E402


def timed(fn, reps=5):
    fn()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


def main():
    be = native.backend()
    dev = "cuda"
    frames = int(os.environ.get("PCS_MB_FRAMES", "12"))
    c1 = make_batch(list(range(frames)))["lidar"].C.to(dev)
    h = be.hash(c1)
    c1 = c1[torch.argsort(h)].contiguous()
    lv = {1: c1}
    for s in (1, 2, 4, 8):
        lv[2 * s] = be.downsample(lv[s], [2 * s] * 3)
    shapes = [(1, 96, 96), (1, 128, 96), (2, 96, 96), (2, 64, 64), (4, 128, 128), (4, 192, 128), (8, 256, 256), (8, 384, 256),
              (16, 256, 256)]
    print("| stride | N | P | cin->cout | fp32 ms | TFLOP/s | bf16 ms | bf16 TFLOP/s | speed-up | GB/s (bf16 algorithmic) | fp16 ms |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    for s, cin, cout in shapes:
        c = lv[s]
        entry = F.build_kernel_map(c, c, (3, 3, 3), (s,) * 3, (1, 1, 1))
        n, p = c.shape[0], entry.fwd.num_pairs
        x = torch.randn(n, cin, device=dev)
        w = torch.randn(27, cin, cout, device=dev) * 0.05
        t32 = timed(lambda: be.conv_gather_gemm(x, w, entry.fwd))
        res = {}
        for dt in (torch.bfloat16, torch.float16):
            wp = be.prepare_weights_h(w, dt, transpose=False)
            xh = x.to(dt)
            res[dt] = timed(lambda: be.conv_gather_gemm_h(xh, wp, 27, cout, entry.fwd))
        fl = 2.0 * p * cin * cout
        by = 2.0 * (n * cin + n * cout) + 8.0 * p + 2.0 * 27 * cin * cout
        tb = res[torch.bfloat16]
        print("| %d | %d | %d | %d->%d | %.3f | %.1f | %.3f | %.1f | %.2fx | %.0f | %.3f |" % (
            s, n, p, cin, cout, t32, fl / t32 / 1e9, tb, fl / tb / 1e9, t32 / tb, by / tb / 1e6, res[torch.float16]))


if __name__ == "__main__":
    main()
