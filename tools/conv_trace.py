#!/usr/bin/env python3
"""Where a fused-conv launch spends its wall time, per wave: needs the debug library built with -DPCS_TRACE=1
(`bash tools/build_debug_lib.sh trace -DPCS_TRACE=1`, see profiles/round1_conv_pmc.md) through PCS_LIB_PATH.
Usage: PCS_LIB_PATH=$PWD/openpcseg_amd/lib/dbg/trace.so python tools/conv_trace.py <level 0..4> <cin> <cout> [frames] [bf16|fp16]
Phases (wall clock, 10 ns ticks): entry->start = LDS zero-fill + offset list + first operand loads; per group:
loop = channel loop (operand loads + MFMAs), ticket = waiting for the earlier groups to commit, commit = LDS RMW;
other = locate/pair loads between groups; end->exit = final barrier + tile write-back."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from openpcseg_amd import functional as F  # noqa: E402
from openpcseg_amd import native  # noqa: E402
from openpcseg_amd.workloads.synthetic import make_batch  # noqa: E402


def main():
    level, cin, cout = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    frames = int(sys.argv[4]) if len(sys.argv) > 4 else 12
    half = {"bf16": torch.bfloat16, "fp16": torch.float16}.get(sys.argv[5]) if len(sys.argv) > 5 else None
    dev = torch.device("cuda:0")
    coords = make_batch(list(range(frames)))["lidar"].C.to(dev)
    coords = coords[torch.argsort(F.sphash(coords))].contiguous()
    ts = 1
    for _ in range(level):
        coords = F.spdownsample(coords, 2, 2, ts)
        ts *= 2
    entry = F.build_kernel_map(coords, coords, (3, 3, 3), (ts,) * 3, (1, 1, 1))
    be = native.backend()
    x = torch.randn(coords.shape[0], cin, device=dev)
    w = torch.randn(27, cin, cout, device=dev) * 0.05
    if half is None:
        run = lambda: be.conv_gather_gemm(x, w, entry.fwd)
    else:
        xh, wp = x.to(half), be.prepare_weights_h(w, half, transpose=False)
        run = lambda: be.conv_gather_gemm_h(xh, wp, 27, cout, entry.fwd)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    run()
    e1.record()
    torch.cuda.synchronize()
    buf = np.zeros(8192 * 64, dtype=np.int64)
    fn = be.lib.pcs_debug_conv_trace if half is None else be.lib.pcs_debug_convh_trace
    fn.restype, fn.argtypes = ctypes.c_int, [ctypes.c_void_p]
    assert fn(buf.ctypes.data) == 0
    t = buf.reshape(8192, 8, 8)
    # per workgroup: lifetime vs row-block groups processed (a proxy of its pairs): how much of the spread is work
    wgs = [(w[w[:, 7] > 0]) for w in t if (w[:, 7] > 0).any()]
    wlife = np.array([(w[:, 7].max() - w[:, 0].min()) * 0.01 for w in wgs])
    wgrp = np.array([w[:, 6].sum() for w in wgs], dtype=np.float64)
    if len(wgs) > 2 and wgrp.std() > 0:
        A = np.stack([wgrp, np.ones_like(wgrp)], 1)
        coef, *_ = np.linalg.lstsq(A, wlife, rcond=None)
        resid = wlife - A @ coef
        print("workgroups traced %d: lifetime mean %.0f us (p5 %.0f, p95 %.0f, max %.0f); groups/WG mean %.1f (p5 %.0f, p95 %.0f, max %.0f)" %
              (len(wgs), wlife.mean(), *np.percentile(wlife, [5, 95, 100]), wgrp.mean(), *np.percentile(wgrp, [5, 95, 100])))
        print("lifetime ~ %.2f us/group * groups + %.0f us; corr %.3f; residual std %.0f us" %
              (coef[0], coef[1], np.corrcoef(wgrp, wlife)[0, 1], resid.std()))
    t = t[t[:, :, 7] > 0]                      # waves that ran
    tick = 0.01                                 # us
    life = (t[:, 7] - t[:, 0]) * tick
    k0, k1 = t[:, 0].min(), t[:, 7].max()
    print("launch %.0f us by events; traced waves %d; first entry -> last exit %.0f us" % (e0.elapsed_time(e1) * 1e3, len(t), (k1 - k0) * tick))
    tot = life.sum()
    pre = ((t[:, 1] - t[:, 0]) * tick).sum()
    loop, ticket, commit = (t[:, 3] * tick).sum(), (t[:, 4] * tick).sum(), (t[:, 5] * tick).sum()
    main_ = ((t[:, 2] - t[:, 1]) * tick).sum()
    post = ((t[:, 7] - t[:, 2]) * tick).sum()
    print("wave lifetime: mean %.1f us, groups/wave mean %.1f" % (life.mean(), t[:, 6].mean()))
    for name, v in [("prologue (zero-fill, offset list, first loads)", pre), ("channel loops", loop), ("ticket wait", ticket),
                    ("commit", commit), ("between groups", main_ - loop - ticket - commit), ("final barrier + write-back", post)]:
        print("  %-48s %5.1f %%" % (name, 100 * v / tot))
    # occupancy over time: how many traced waves are alive
    span = (k1 - k0) * tick
    alive = life.sum() / span
    print("mean traced waves alive %.0f (of %d wave slots at 2 workgroups/CU)" % (alive, 256 * 8))
    # timeline: waves alive in 20 equal time bins (a long thin tail = load imbalance between workgroups)
    edges = np.linspace(k0, k1, 21)
    occ = [float(np.clip(np.minimum(t[:, 7], hi) - np.maximum(t[:, 0], lo), 0, None).sum() / (hi - lo)) for lo, hi in zip(edges[:-1], edges[1:])]
    print("alive per 5 % time bin:", " ".join("%4.0f" % o for o in occ))
    wg = t[:, 7] - t[:, 0]
    print("wave lifetime percentiles (us): p10 %.0f p50 %.0f p90 %.0f max %.0f" % tuple(np.percentile(wg, [10, 50, 90, 100]) * tick))


if __name__ == "__main__":
    main()
