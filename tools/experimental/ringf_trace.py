#!/usr/bin/env python3
"""Phase timers of conv_ring6f_kernel (variant library built with -DPCS_RING_TRACE=1, selected through PCS_LIB_PATH):
per wave of the first workgroups: prologue / main loop / epilogue cycles, cycles inside s_barrier, the loader's DMA-issue and
DMA-wait cycles.   python tools/ringf_trace.py "<level> <cin> <cout>" ..."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from openpcseg_amd import functional as F  # noqa: E402
from openpcseg_amd import native  # noqa: E402
from openpcseg_amd.workloads.synthetic import make_batch  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    coords = make_batch(list(range(int(os.environ.get("PCS_AB_FRAMES", "12")))))["lidar"].C.to(dev)
    coords = coords[torch.argsort(F.sphash(coords))].contiguous()
    levels, ts = [coords], 1
    for _ in range(4):
        levels.append(F.spdownsample(levels[-1], 2, 2, ts))
        ts *= 2
    be = native.backend()
    be.lib.pcs_conv_ring_enable(0, 1)
    fn = be.lib.pcs_debug_ringf_trace
    fn.restype, fn.argtypes = ctypes.c_int, [ctypes.c_void_p]
    for spec in sys.argv[1:]:
        level, cin, cout = (int(v) for v in spec.split())
        c = levels[level]
        entry = F.build_kernel_map(c, c, (3, 3, 3), (2 ** level,) * 3, (1, 1, 1))
        x = torch.randn(c.shape[0], cin, device=dev)
        w = torch.randn(27, cin, cout, device=dev) * 0.05
        for _ in range(3):
            be.conv_gather_gemm(x, w, entry.fwd)
        torch.cuda.synchronize()
        buf = np.zeros((1024, 8, 8), dtype=np.int64)
        assert fn(buf.ctypes.data) == 0
        t = be.tile_rows(cin, cout, entry.fwd)
        used = buf[:, :, 3] > 0
        nw = int(used[0].sum())
        b = buf[used[:, 0]]
        print("level %d %d->%d tile %d: %d workgroups traced, %d waves each (last = loader)" % (level, cin, cout, t, len(b), nw))
        for w_ in range(nw):
            r = b[:, w_, :].astype(np.float64)
            total = r[:, 3] - r[:, 0]
            print("  wave %d: total %7.0f  prologue %6.0f  main %7.0f  epilogue %6.0f | in barrier %7.0f (%4.1f %% of main)  "
                  "wait %7.0f  issue %7.0f  batches %5.1f  -> %6.0f cycles / batch" % (
                      w_, total.mean(), (r[:, 1] - r[:, 0]).mean(), (r[:, 2] - r[:, 1]).mean(), (r[:, 3] - r[:, 2]).mean(),
                      r[:, 4].mean(), 100 * r[:, 4].mean() / max((r[:, 2] - r[:, 1]).mean(), 1), r[:, 5].mean(), r[:, 6].mean(),
                      r[:, 7].mean(), (r[:, 2] - r[:, 1]).mean() / max(r[:, 7].mean(), 1)))


if __name__ == "__main__":
    main()
