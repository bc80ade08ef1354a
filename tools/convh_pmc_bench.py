#!/usr/bin/env python3
"""The two reference shapes of the half (bf16) fused convolution, a few launches each, for counter collection:
stride 1 96->96 (conv_os5h_kernel<Bf16, 6, ...>) and stride 8 256->256 (conv_os5h_kernel<Bf16, 8, ...>) on the 12-frame
bench maps. Prints the HIP-event time per launch; run under rocprofv3 --pmc by tools/convh_pmc.sh."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from openpcseg_amd import functional as F  # noqa: E402
from openpcseg_amd import native  # noqa: E402
from openpcseg_amd.workloads.synthetic import make_batch  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    dev = torch.device("cuda:0")
    coords = make_batch(list(range(12)))["lidar"].C.to(dev)
    coords = coords[torch.argsort(F.sphash(coords))].contiguous()
    levels, ts = [coords], 1
    for _ in range(3):
        levels.append(F.spdownsample(levels[-1], 2, 2, ts))
        ts *= 2
    be = native.backend()
    shapes = ((0, 96, 96), (3, 256, 256))
    if os.environ.get("PCS_PMC_SHAPES"):   # "level cin cout;level cin cout"
        shapes = tuple(tuple(int(v) for v in sp.split()) for sp in os.environ["PCS_PMC_SHAPES"].split(";"))
    for level, cin, cout in shapes:
        c = levels[level]
        entry = F.build_kernel_map(c, c, (3, 3, 3), (2 ** level,) * 3, (1, 1, 1))
        n, p = c.shape[0], entry.fwd.num_pairs
        x = torch.randn(n, cin, device=dev).to(torch.bfloat16)
        wp = be.prepare_weights_h(torch.randn(27, cin, cout, device=dev) * 0.05, torch.bfloat16, transpose=False)
        be.conv_gather_gemm_h(x, wp, 27, cout, entry.fwd)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            be.conv_gather_gemm_h(x, wp, 27, cout, entry.fwd)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        alg = 2.0 * (n * cin + n * cout) + 8.0 * p + 2.0 * 27 * cin * cout
        print("convh level=%d n=%d pairs=%d %d->%d tile=%d: %.0f us  %.1f TFLOP/s  algorithmic %.1f MB = %.0f GB/s" %
              (level, n, p, cin, cout, be.tile_rows(cin, cout, entry.fwd, dtype=1),
               us, 2.0 * p * cin * cout / us / 1e6, alg / 1e6, alg / us / 1e3), flush=True)


if __name__ == "__main__":
    main()
