#!/usr/bin/env python3
"""Tile-height sweep over every k3 fused-conv shape of MinkUNet-34 cr1.0 on the 12-frame bench maps: the data behind
pcs_conv_pick_tile_rows. Usage: python tools/conv_shape_sweep.py [f32|bf16] [tile,tile,...]  -> markdown table."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from openpcseg_amd import functional as F  # noqa: E402
from openpcseg_amd import native  # noqa: E402
from openpcseg_amd.workloads.synthetic import make_batch  # noqa: E402

SHAPES = [(0, 96, 96), (0, 128, 96), (0, 96, 128), (1, 96, 96), (1, 128, 96), (1, 96, 128), (2, 128, 128), (2, 64, 64),
          (2, 192, 128), (2, 128, 192), (2, 32, 64), (2, 64, 32), (3, 256, 256), (3, 128, 128), (3, 384, 256), (3, 256, 384),
          (3, 64, 128), (3, 128, 64), (4, 256, 256), (4, 128, 256), (4, 256, 128)]


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "f32"
    tiles = [int(t) for t in sys.argv[2].split(",")] if len(sys.argv) > 2 else [96, 112, 128, 144, 160, 176, 192, 208, 224, 256, 288, 320, 384]
    reps = int(os.environ.get("PCS_SWEEP_REPS", "40"))
    dev = torch.device("cuda:0")
    coords = make_batch(list(range(12)))["lidar"].C.to(dev)
    coords = coords[torch.argsort(F.sphash(coords))].contiguous()
    levels, ts = [coords], 1
    for _ in range(4):
        levels.append(F.spdownsample(levels[-1], 2, 2, ts))
        ts *= 2
    be = native.backend()
    maps = {}
    half = {"bf16": torch.bfloat16, "fp16": torch.float16}.get(mode)
    print("| level | cin->cout | picked | " + " | ".join(str(t) for t in tiles) + " | best |")
    print("|---|---|---|" + "---|" * (len(tiles) + 1))
    warm = 300
    for level, cin, cout in SHAPES:
        c = levels[level]
        if level not in maps:
            maps[level] = F.build_kernel_map(c, c, (3, 3, 3), (2 ** level,) * 3, (1, 1, 1))
        kmap = maps[level].fwd
        x = torch.randn(c.shape[0], cin, device=dev)
        w = torch.randn(27, cin, cout, device=dev) * 0.05
        if half is not None:
            if not be.conv_h_applies(cin, cout, 27):
                continue
            xh, wp = x.to(half), be.prepare_weights_h(w, half, transpose=False)
        res = {}
        for t in [0] + tiles:
            tt = t or None
            run = (lambda: be.conv_gather_gemm(x, w, kmap, tile_rows=tt)) if half is None else \
                  (lambda: be.conv_gather_gemm_h(xh, wp, 27, cout, kmap, tile_rows=tt))
            try:
                for _ in range(warm):
                    run()
            except RuntimeError:
                res[t] = None
                continue
            warm = 10
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                run()
            e1.record()
            torch.cuda.synchronize()
            res[t] = e0.elapsed_time(e1) * 1e3 / reps
        best = min((v, k) for k, v in res.items() if v is not None and k)
        print("| %d | %d->%d | %d: %.0f | %s | %d: %.0f (%.1f TF) |" % (
            level, cin, cout, be.tile_rows(cin, cout, kmap), res[0],
            " | ".join("-" if res[t] is None else "%.0f" % res[t] for t in tiles), best[1], best[0],
            2.0 * kmap.num_pairs * cin * cout / best[0] / 1e6), flush=True)


if __name__ == "__main__":
    main()
