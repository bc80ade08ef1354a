#!/usr/bin/env python3
"""Fused-conv time over tile heights on the real rulebooks of the synthetic 12-frame batch.
Usage: python tools/conv_tile_sweep.py "<level> <cin> <cout> <tile,tile,...> [bf16|fp16]" ...   (tile 0 = the picker's choice;
<tile>+rows = row order instead of the heaviest-first tile order)
Set PCS_CONV_DEBUG=1 to see the kernel instance and the resident workgroups per CU of every configuration."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from openpcseg_amd import functional as F  # noqa: E402
from openpcseg_amd import native  # noqa: E402
from openpcseg_amd.workloads.synthetic import make_batch  # noqa: E402


def main():
    frames = int(os.environ.get("PCS_SWEEP_FRAMES", "12"))
    reps = int(os.environ.get("PCS_SWEEP_REPS", "100"))
    dev = torch.device("cuda:0")
    coords = make_batch(list(range(frames)))["lidar"].C.to(dev)
    if os.environ.get("PCS_SWEEP_SORT0", "hash") == "hash":   # "ravel": level 0 as the input pipeline delivers it
        coords = coords[torch.argsort(F.sphash(coords))].contiguous()
    levels, ts = [coords], 1
    for _ in range(4):
        levels.append(F.spdownsample(levels[-1], 2, 2, ts))
        ts *= 2
    be = native.backend()
    maps = {}
    for spec in sys.argv[1:]:
        level, cin, cout, tiles = spec.split()[:4]
        half = {"bf16": torch.bfloat16, "fp16": torch.float16}.get(spec.split()[4]) if len(spec.split()) > 4 else None
        level, cin, cout = int(level), int(cin), int(cout)
        c = levels[level]
        if level not in maps:
            maps[level] = F.build_kernel_map(c, c, (3, 3, 3), (2 ** level,) * 3, (1, 1, 1))
        entry = maps[level]
        n, p = c.shape[0], entry.fwd.num_pairs
        x = torch.randn(n, cin, device=dev)
        w = torch.randn(27, cin, cout, device=dev) * 0.05
        if half is not None:
            xh, wp = x.to(half), be.prepare_weights_h(w, half, transpose=False)
        ref = None
        warm = 300
        for tile in tiles.split(","):
            opts = tile.split("+")[1:]          # e.g. 224+rows = row order (no heaviest-first tile order)
            tile = int(tile.split("+")[0])
            t = tile or None
            ordered = "rows" not in opts
            if half is None:
                run = lambda: be.conv_gather_gemm(x, w, entry.fwd, tile_rows=t, ordered=ordered)
            else:
                run = lambda: be.conv_gather_gemm_h(xh, wp, 27, cout, entry.fwd, tile_rows=t, ordered=ordered)
            y = run().float()
            if ref is None:
                ref = y
            err = float((y - ref).abs().max() / ref.abs().max())
            for _ in range(warm):
                run()
            warm = 20
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                run()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / reps
            print("level=%d n=%d pairs=%d %d->%d tile=%s%s%s: %.0f us  %.1f TFLOP/s  (diff vs first %.1e)" %
                  (level, n, p, cin, cout, tile or be.tile_rows(cin, cout, entry.fwd, 0 if half is None else be._HALF[half]), "".join("+" + o for o in opts), (" " + spec.split()[4]) if half is not None else "", us,
                   2.0 * p * cin * cout / us / 1e6, err), flush=True)


if __name__ == "__main__":
    main()
