#!/usr/bin/env python3
"""Small-map check of the ring kernel against the fp32 MFMA convolution of the same (half-rounded) operands, with a
breakdown of where the mismatches sit (row tile, row inside the tile, 16-column tile) when there are any.
python tools/ring_debug.py [n_points]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from openpcseg_amd import functional as F  # noqa: E402
from openpcseg_amd import native  # noqa: E402
from openpcseg_amd.workloads.synthetic import make_batch  # noqa: E402


def main():
    npts = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
    dev = torch.device("cuda:0")
    be = native.backend()
    be.lib.pcs_conv_ring_enable(1, 1)
    coords = make_batch([0, 1], n_points=npts)["lidar"].C.to(dev)
    coords = coords[torch.argsort(F.sphash(coords))].contiguous()
    entry = F.build_kernel_map(coords, coords, (3, 3, 3), (1, 1, 1), (1, 1, 1))
    n = coords.shape[0]
    print("n=%d pairs=%d" % (n, entry.fwd.num_pairs), flush=True)
    be.lib.pcs_conv_ring_enable(0, 1)
    for cin, cout, tile in [(64, 128, None), (32, 64, None), (96, 96, None), (128, 96, 48), (256, 256, None), (64, 64, 96)]:  # fp32 ring vs the wave kernel
        torch.manual_seed(cin * 5 + cout)
        x = torch.randn(n, cin, device=dev)
        w = torch.randn(27, cin, cout, device=dev) * 0.05
        assert be.lib.pcs_conv_ring_applies(cin, cout, 27, tile or be.tile_rows(cin, cout, entry.fwd), 0) == 1
        y = be.conv_gather_gemm(x, w, entry.fwd, tile_rows=tile)
        be.lib.pcs_conv_ring_enable(0, 0)
        ref = be.conv_gather_gemm(x, w, entry.fwd)
        be.lib.pcs_conv_ring_enable(0, 1)
        torch.cuda.synchronize()
        d = (y - ref).abs()
        bad = d > 2e-5 * ref.abs().max()
        t = tile or be.tile_rows(cin, cout, entry.fwd)
        print("f32 %d->%d tile=%d max_err=%.3e (ref max %.2f) bad=%d of %d" % (cin, cout, t, float(d.max()), float(ref.abs().max()),
                                                                          int(bad.sum()), bad.numel()), flush=True)
        if bad.any():
            b = bad.cpu().numpy()
            rows, cols = np.nonzero(b)
            print("  bad rows: %d distinct; by row tile %s; by 16-column tile %s" % (len(set(rows.tolist())), np.bincount(rows // t)[:12].tolist(),
                                                                                    np.bincount(cols // 16, minlength=cout // 16).tolist()))
    be.lib.pcs_conv_ring_enable(0, -1)
    for cin, cout, tile in [(128, 128, 64), (128, 128, None), (96, 96, None), (64, 128, None), (256, 128, None), (192, 96, 96),
                            (384, 256, None), (128, 96, 48)]:
        torch.manual_seed(cin * 7 + cout)
        x = torch.randn(n, cin, device=dev).to(torch.bfloat16)
        w = (torch.randn(27, cin, cout, device=dev) * 0.05).to(torch.bfloat16).float()
        wp = be.prepare_weights_h(w, torch.bfloat16, transpose=False)
        ref = be.conv_gather_gemm(x.float(), w, entry.fwd)
        y = be.conv_gather_gemm_h(x, wp, 27, cout, entry.fwd, tile_rows=tile).float()
        torch.cuda.synchronize()
        t = tile or be.tile_rows(cin, cout, entry.fwd, 1)
        d = (y - ref).abs()
        tol = 1.0 / 128 * ref.abs() + 1e-4 * ref.abs().max()
        bad = d > tol
        print("%d->%d tile=%d max_err=%.3e (ref max %.2f) bad=%d of %d" % (cin, cout, t, float(d.max()), float(ref.abs().max()),
                                                                       int(bad.sum()), bad.numel()), flush=True)
        if bad.any():
            b = bad.cpu().numpy()
            rows, cols = np.nonzero(b)
            print("  bad rows: %d distinct; first %s" % (len(set(rows.tolist())), sorted(set(rows.tolist()))[:12]))
            print("  by row tile:", np.bincount(rows // t)[:16].tolist())
            print("  by row in tile (16-row bins):", np.bincount((rows % t) // 16).tolist())
            print("  by 16-column tile:", np.bincount(cols // 16, minlength=cout // 16).tolist())
            r0 = rows[0]
            print("  row %d: y %s ref %s" % (r0, y[r0, :6].cpu().numpy().round(3).tolist(), ref[r0, :6].cpu().numpy().round(3).tolist()))


if __name__ == "__main__":
    main()
