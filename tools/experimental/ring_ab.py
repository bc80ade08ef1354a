#!/usr/bin/env python3
"""[needs the variant library: bash tools/build_variant_lib.sh ring; PCS_LIB_PATH=openpcseg_amd/lib/dbg/ring.so]
A/B of the two fused-convolution structures on the real rulebooks of the synthetic batch: the wave-autonomous kernels
(conv_os5_kernel fp32 / conv_os5h_kernel bf16) against the column-parallel ring kernels (conv_ring6f.hip / conv_ring6h.hip),
switched at run time through pcs_conv_ring_enable. One line per (dtype, shape): time and TFLOP/s of both, the tile heights the
picker chose, and the largest difference between the two results relative to the tensor maximum.

  python tools/ring_ab.py [f32|bf16|both]
  PCS_AB_FRAMES (12), PCS_AB_REPS (30), PCS_AB_SHAPES ("level cin cout;...")."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

DEFAULT_SHAPES = "0 96 96;0 128 96;1 96 96;1 128 96;2 128 128;2 192 128;2 64 64;3 256 256;3 384 256;3 128 128;4 256 256;1 32 64"


def main():
    import torch
    from openpcseg_amd import functional as F
    from openpcseg_amd import native
    from openpcseg_amd.workloads.synthetic import make_batch
    which = sys.argv[1] if len(sys.argv) > 1 else "both"
    frames = int(os.environ.get("PCS_AB_FRAMES", "12"))
    reps = int(os.environ.get("PCS_AB_REPS", "30"))
    shapes = [tuple(int(v) for v in s.split()) for s in os.environ.get("PCS_AB_SHAPES", DEFAULT_SHAPES).split(";") if s.strip()]
    dev = torch.device("cuda:0")
    coords = make_batch(list(range(frames)))["lidar"].C.to(dev)
    coords = coords[torch.argsort(F.sphash(coords))].contiguous()
    levels, ts = [coords], 1
    for _ in range(4):
        levels.append(F.spdownsample(levels[-1], 2, 2, ts))
        ts *= 2
    be = native.backend()
    lib = be.lib
    maps = {}

    def timed(run):
        for _ in range(8):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            run()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / reps

    print("| dtype | level | cin->cout | wave us (tile) TFLOP/s | ring us (tile) TFLOP/s | ring / wave | max diff |")
    print("|---|---|---|---|---|---|---|", flush=True)
    for dt in (["f32", "bf16"] if which == "both" else [which]):
        kind, code = (0, 0) if dt == "f32" else (1, 1)
        for level, cin, cout in shapes:
            c = levels[level]
            if level not in maps:
                maps[level] = F.build_kernel_map(c, c, (3, 3, 3), (2 ** level,) * 3, (1, 1, 1))
            entry = maps[level]
            p = entry.fwd.num_pairs
            torch.manual_seed(level * 1000 + cin + cout)
            x = torch.randn(c.shape[0], cin, device=dev)
            w = torch.randn(27, cin, cout, device=dev) * 0.05
            if dt == "f32":
                run = lambda: be.conv_gather_gemm(x, w, entry.fwd)
            else:
                xh, wp = x.to(torch.bfloat16), be.prepare_weights_h(w, torch.bfloat16, transpose=False)
                run = lambda: be.conv_gather_gemm_h(xh, wp, 27, cout, entry.fwd)
            res = {}
            for mode in (0, 1):
                lib.pcs_conv_ring_enable(kind, mode)
                t = be.tile_rows(cin, cout, entry.fwd, code)
                ring = bool(lib.pcs_conv_ring_applies(cin, cout, 27, t, code))
                if mode == 1 and not ring:
                    res[mode] = None
                    continue
                y = run().float()
                res[mode] = (timed(run), t, y)
            lib.pcs_conv_ring_enable(kind, -1)
            us0, t0, y0 = res[0]
            fl = 2.0 * p * cin * cout
            if res[1] is None:
                print("| %s | %d | %d->%d | %.1f (%d) %.1f | - | - | - |" % (dt, level, cin, cout, us0, t0, fl / us0 / 1e6), flush=True)
                continue
            us1, t1, y1 = res[1]
            diff = float((y1 - y0).abs().max() / y0.abs().max())
            print("| %s | %d | %d->%d | %.1f (%d) %.1f | %.1f (%d) %.1f | %.2fx | %.1e |" % (
                dt, level, cin, cout, us0, t0, fl / us0 / 1e6, us1, t1, fl / us1 / 1e6, us0 / us1, diff), flush=True)


if __name__ == "__main__":
    main()
