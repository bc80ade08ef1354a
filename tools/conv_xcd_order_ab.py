#!/usr/bin/env python3
"""Launch order of the fused convolution's row tiles vs the XCD a workgroup lands on (workgroup id % 8), on the bench's own maps
(deeper levels in spdownsample's coordinate order, as in the network; level 0 in coordinate order by default -- inside the network
the stride-1 voxels are in ascending-hash order, initial_voxelize: PCS_AB_SORT0=hash). Orders: rows (tile = workgroup id), heavy (heaviest first, the product's choice on dense levels), xcd (every XCD walks a
contiguous eighth of the row order), xcdheavy (contiguous eighths, heaviest first inside each).
Usage: python tools/conv_xcd_order_ab.py "<level> <cin> <cout> [bf16]" ...   PCS_AB_SORT0=hash puts level 0 in hash order."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from openpcseg_amd import functional as F  # noqa: E402
from openpcseg_amd import native  # noqa: E402
from openpcseg_amd.workloads.synthetic import make_batch  # noqa: E402


def xcd_contiguous(seq, nx=8):
    """seq (1-D tensor of tile ids in the order one XCD should walk them, concatenated over XCDs) -> launch order in which
    position p (XCD p % nx) takes the next tile of chunk p % nx."""
    n = seq.numel()
    chunk = (n + nx - 1) // nx
    pad = torch.full((chunk * nx,), -1, dtype=seq.dtype, device=seq.device)
    pad[:n] = seq
    out = pad.view(nx, chunk).t().reshape(-1)
    return out[out >= 0].contiguous()


def main():
    frames = int(os.environ.get("PCS_SWEEP_FRAMES", "12"))
    reps = int(os.environ.get("PCS_SWEEP_REPS", "60"))
    dev = torch.device("cuda:0")
    coords = make_batch(list(range(frames)))["lidar"].C.to(dev)
    if os.environ.get("PCS_AB_SORT0") == "hash":
        coords = coords[torch.argsort(F.sphash(coords))].contiguous()
    levels, ts = [coords], 1
    for _ in range(4):
        levels.append(F.spdownsample(levels[-1], 2, 2, ts))
        ts *= 2
    be = native.backend()
    maps = {}
    for spec in sys.argv[1:]:
        f = spec.split()
        level, cin, cout = int(f[0]), int(f[1]), int(f[2])
        half = {"bf16": torch.bfloat16, "fp16": torch.float16}.get(f[3]) if len(f) > 3 else None
        c = levels[level]
        if level not in maps:
            maps[level] = F.build_kernel_map(c, c, (3, 3, 3), (2 ** level,) * 3, (1, 1, 1))
        kmap = maps[level].fwd
        n, p = c.shape[0], kmap.num_pairs
        x = torch.randn(n, cin, device=dev)
        w = torch.randn(27, cin, cout, device=dev) * 0.05
        if half is not None:
            xh, wp = x.to(half), be.prepare_weights_h(w, half, transpose=False)
        t = be.tile_rows(cin, cout, kmap, 0 if half is None else be._HALF[half])
        ntiles = (n + t - 1) // t
        kmap._seg.pop(("order", t), None)
        heavy = be._tile_order(kmap, t).clone()
        rows = torch.arange(ntiles, dtype=torch.int32, device=dev)
        chunk = (ntiles + 7) // 8
        # heaviest first inside each contiguous eighth: stable partition of the heavy order by chunk id
        part = torch.cat([heavy[(heavy // chunk) == k] for k in range(8)])
        orders = {"rows": None, "heavy": heavy, "xcd": xcd_contiguous(rows), "xcdheavy": xcd_contiguous(part)}
        # xb<BS>: contiguous eighths (balanced lengths), inside each the blocks of BS consecutive tiles heaviest block first, row
        # order inside a block (the tiles in flight on an XCD stay neighbours, the light ones still come last)
        seg = be._segments(kmap, t).view(kmap.K, ntiles + 1).long()
        work = ((seg[:, 1:] - seg[:, :-1] + 15) // 16).sum(0)
        q, r = divmod(ntiles, 8)
        for bs in (16, 32, 64):
            seqs = []
            for c in range(8):
                lo, ln = c * q + min(c, r), q + (1 if c < r else 0)
                tl = torch.arange(lo, lo + ln, device=dev)
                nb = (ln + bs - 1) // bs
                bw = torch.zeros(max(nb, 1), dtype=torch.long, device=dev).index_add_(0, (tl - lo) // bs, work[lo:lo + ln])
                rank = torch.argsort(torch.argsort(-bw[:nb], stable=True), stable=True)      # block -> place
                key = rank[(tl - lo) // bs] * bs + (tl - lo) % bs
                seqs.append(tl[torch.argsort(key, stable=True)])
            out = torch.full((8 * (q + 1),), -1, dtype=torch.long, device=dev)
            for c in range(8):
                out[c:c + 8 * seqs[c].numel():8] = seqs[c]
            orders["xb%d" % bs] = out[out >= 0].to(torch.int32)
        ref, base = None, None
        only = os.environ.get("PCS_AB_ONLY")   # one order only (counter runs: every conv launch of the process is that order)
        for name, order in orders.items():
            if only and name != only:
                continue
            if order is None:
                kw = dict(ordered=False)
            else:
                assert order.numel() == ntiles and int(torch.sort(order.long())[0].ne(rows.long()).sum()) == 0
                kmap._seg[("order", t)] = order.to(torch.int32).contiguous()
                kw = dict(ordered="force")
            if half is None:
                run = lambda: be.conv_gather_gemm(x, w, kmap, tile_rows=t, **kw)
            else:
                run = lambda: be.conv_gather_gemm_h(xh, wp, 27, cout, kmap, tile_rows=t, **kw)
            y = run().float()
            ref = y if ref is None else ref
            same = bool(torch.equal(y, ref))
            for _ in range(30):
                run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                run()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / reps
            base = us if base is None else base
            print("level=%d n=%d pairs=%d %d->%d %s tile=%d %-8s: %6.0f us  %6.1f TFLOP/s  x%.2f  identical %s" %
                  (level, n, p, cin, cout, f[3] if half is not None else "fp32", t, name, us, 2.0 * p * cin * cout / us / 1e6,
                   base / us, same), flush=True)
        kmap._seg.pop(("order", t), None)


if __name__ == "__main__":
    main()
