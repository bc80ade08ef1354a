#!/usr/bin/env python3
"""A/B of the launch order of the weight-gradient splits (csrc/conv_wgrad.hip::find_split_wave): offset-major (round 2-5) vs
position-major (round 6), bf16 (wgrad3) and fp32 (wgrad2) operands on the 12-frame bench maps; results must be bit-identical."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from openpcseg_amd import functional as F  # noqa: E402
from openpcseg_amd import native  # noqa: E402
from openpcseg_amd.workloads.synthetic import make_batch  # noqa: E402


def timeit(run, reps=30, warm=10):
    for _ in range(warm):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    be = native.backend()
    mode = be.lib.pcs_debug_wgrad_interleave
    mode.restype, mode.argtypes = None, [ctypes.c_int32]
    dev = "cuda"
    c1 = make_batch(list(range(12)))["lidar"].C.to(dev)
    c1 = c1[torch.argsort(be.hash(c1))].contiguous()
    lv = {1: c1}
    for s in (1, 2, 4, 8):
        lv[2 * s] = be.downsample(lv[s], [2 * s] * 3)
    shapes = [(1, 32, 32), (1, 96, 96), (1, 128, 96), (2, 64, 64), (2, 96, 96), (4, 128, 128), (4, 192, 128), (8, 256, 256),
              (8, 384, 256), (16, 256, 256)]
    tot = {}
    for s, cin, cout in shapes:
        c = lv[s]
        entry = F.build_kernel_map(c, c, (3, 3, 3), (s,) * 3, (1, 1, 1))
        p = entry.fwd.num_pairs
        x = torch.randn(c.shape[0], cin, device=dev)
        gy = torch.randn(c.shape[0], cout, device=dev)
        xh, gh = x.bfloat16(), gy.bfloat16()
        fl = 2.0 * p * cin * cout
        row = []
        for name, run in (("bf16", lambda: be.conv_wgrad_h(xh, gh, entry.fwd, 0)), ("fp32", lambda: be.conv_wgrad(x, gy, entry.fwd, 0))):
            mode(0)
            r0 = run()
            t0 = timeit(run)
            mode(1)
            r1 = run()
            t1 = timeit(run)
            mode(2)
            r2 = run()
            t2 = timeit(run)
            mode(-1)
            tot[name] = tuple(a + b for a, b in zip(tot.get(name, (0.0, 0.0, 0.0)), (t0, t1, t2)))
            row.append("%s %.0f -> %.0f / xcd %.0f us x%.2f / x%.2f %s" % (name, t0, t1, t2, t0 / t1, t0 / t2,
                                                                        "bit-identical" if torch.equal(r0, r1) and torch.equal(r0, r2) else "DIFFERENT"))
        print("stride %2d %3d x %3d P=%d: %s" % (s, cin, cout, p, " | ".join(row)), flush=True)
    for k, v in tot.items():
        print("total %s: %.2f -> %.2f / xcd %.2f ms" % (k, v[0] / 1e3, v[1] / 1e3, v[2] / 1e3))


if __name__ == "__main__":
    main()
