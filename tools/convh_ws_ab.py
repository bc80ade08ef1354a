#!/usr/bin/env python3
"""A/B of the weight-stationary half convolution (conv_wave6h.hip) against conv_wave5h.hip on the bench maps: time and
result difference per (shape, tile height, row blocks per unit).
python tools/convh_ws_ab.py "<level> <cin> <cout> <tile,tile,...> [mode[,mode]] [rs] [s|d|t]" ...     (one line per configuration;
mode 1 = the kernel's shape policy, 2 = every shape it serves; s = k3 submanifold map of the level (default), d = k2 s2 map
level -> level + 1, t = its transposed use level + 1 -> level)"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from openpcseg_amd import functional as F  # noqa: E402
from openpcseg_amd import native  # noqa: E402
from openpcseg_amd.workloads.synthetic import make_batch  # noqa: E402


def timeit(run, reps, warm):
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
    frames = int(os.environ.get("PCS_SWEEP_FRAMES", "12"))
    reps = int(os.environ.get("PCS_SWEEP_REPS", "40"))
    dtype = {"bf16": torch.bfloat16, "fp16": torch.float16}[os.environ.get("PCS_AB_DTYPE", "bf16")]
    dev = torch.device("cuda:0")
    coords = make_batch(list(range(frames)))["lidar"].C.to(dev)
    coords = coords[torch.argsort(F.sphash(coords))].contiguous()
    levels, ts = [coords], 1
    for _ in range(4):
        levels.append(F.spdownsample(levels[-1], 2, 2, ts))
        ts *= 2
    be = native.backend()
    ws = be.lib.pcs_debug_convh_ws
    ws.restype = None
    ws.argtypes = [ctypes.c_int32] * 3
    maps = {}
    warm = 300
    for spec in sys.argv[1:]:
        f = spec.split()
        level, cin, cout = int(f[0]), int(f[1]), int(f[2])
        tiles = [int(t) for t in f[3].split(",")]
        rus = [int(t) for t in f[4].split(",")] if len(f) > 4 else [4]
        rs = int(f[5]) if len(f) > 5 else 0
        kind = f[6] if len(f) > 6 else "s"
        c = levels[level]
        key = (level, kind != "s")
        if key not in maps:
            maps[key] = (F.build_kernel_map(c, c, (3, 3, 3), (2 ** level,) * 3, (1, 1, 1)) if kind == "s" else
                         F.build_kernel_map(c, levels[level + 1], (2, 2, 2), (2 ** level,) * 3, (1, 1, 1)))
        entry = maps[key]
        kmap = entry.rev if kind == "t" else entry.fwd
        kv = kmap.K
        if kind == "d":
            c = levels[level + 1]
        nsrc = levels[level + 1].shape[0] if kind == "t" else levels[level].shape[0]
        n, p = c.shape[0], kmap.num_pairs
        torch.manual_seed(0)
        x = torch.randn(nsrc, cin, device=dev).to(dtype)
        w = torch.randn(kv, cin, cout, device=dev) * 0.05
        wp = be.prepare_weights_h(w, dtype, transpose=False)
        flops = 2.0 * p * cin * cout
        ws(0, 0, 0)
        run0 = lambda: be.conv_gather_gemm_h(x, wp, kv, cout, kmap)
        ref = run0().float()
        t0 = timeit(run0, reps, warm)
        warm = 20
        print("L%d%s %d->%d n=%d P=%d  wave5h picker tile=%d: %.0f us %.1f TF/s" % (
            level, kind, cin, cout, n, p, be.tile_rows(cin, cout, kmap, dtype=1), t0, flops / t0 / 1e6), flush=True)
        scale = float(ref.abs().max())
        for tile in tiles:
            for ru in rus:
                ws(ru if ru > 0 else 2, 0, rs)
                run = lambda: be.conv_gather_gemm_h(x, wp, kv, cout, kmap, tile_rows=tile or None)
                y = run().float()
                y2 = run().float()
                err = float((y - ref).abs().max()) / scale
                nbad = int(((y - ref).abs() > 0.02 * scale).sum())
                rep = bool((y == y2).all())
                t1 = timeit(run, reps, 20)
                print("   ws tile=%d mode=%d rs=%d: %.0f us %.1f TF/s  x%.2f  maxdiff/scale %.1e  bad %d  bit-repro %s" % (
                    tile, ru, rs, t1, flops / t1 / 1e6, t0 / t1, err, nbad, rep), flush=True)
        ws(-1, 0, 0)


if __name__ == "__main__":
    main()
