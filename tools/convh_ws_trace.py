#!/usr/bin/env python3
"""Per-wave phase timers of the weight-stationary half convolution (library built with -DPCS_TRACE=1 or 2, see
tools/build_debug_convh.sh; select it with PCS_LIB_PATH).  python tools/convh_ws_trace.py <level> <cin> <cout> <tile> [ru] [rs]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from openpcseg_amd import functional as F  # noqa: E402
from openpcseg_amd import native  # noqa: E402
from openpcseg_amd.workloads.synthetic import make_batch  # noqa: E402


def main():
    level, cin, cout, tile = (int(v) for v in sys.argv[1:5])
    ru = int(sys.argv[5]) if len(sys.argv) > 5 else -1
    rs = int(sys.argv[6]) if len(sys.argv) > 6 else 0
    dev = torch.device("cuda:0")
    coords = make_batch(list(range(12)))["lidar"].C.to(dev)
    coords = coords[torch.argsort(F.sphash(coords))].contiguous()
    ts = 1
    for _ in range(level):
        coords = F.spdownsample(coords, 2, 2, ts)
        ts *= 2
    be = native.backend()
    be.lib.pcs_debug_convh_ws.restype = None
    be.lib.pcs_debug_convh_ws.argtypes = [ctypes.c_int32] * 3
    be.lib.pcs_debug_ws_trace.argtypes = [ctypes.c_void_p, ctypes.c_int32]
    entry = F.build_kernel_map(coords, coords, (3, 3, 3), (ts,) * 3, (1, 1, 1))
    n, p = coords.shape[0], entry.fwd.num_pairs
    x = torch.randn(n, cin, device=dev).to(torch.bfloat16)
    wp = be.prepare_weights_h(torch.randn(27, cin, cout, device=dev) * 0.05, torch.bfloat16, transpose=False)
    be.lib.pcs_debug_convh_ws(1, ru, rs)
    run = lambda: be.conv_gather_gemm_h(x, wp, 27, cout, entry.fwd, tile_rows=tile)
    for _ in range(50):
        run()
    nb = 16384
    buf = torch.zeros(nb * 64, dtype=torch.int64, device=dev)
    be.lib.pcs_debug_ws_trace(buf.data_ptr(), nb)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record()
    torch.cuda.synchronize()
    be.lib.pcs_debug_ws_trace(None, 0)
    t = buf.view(nb, 8, 8).cpu()
    live = t[:, :, 0] != 0
    t = t[live].double()
    nsub = (t[:, 7].long() & 255).double()
    t_exit = (t[:, 7].long() >> 8).double()
    opw = (t[:, 5].long() >> 32).double()
    tick = (t[:, 5].long() & 0xFFFFFFFF).double()
    life = t_exit - t[:, 0]
    tot = life.sum()
    print("L%d %d->%d tile=%d ru=%d rs=%d: launch %.0f us; waves traced %d, sub-groups/wave %.1f, wave lifetime %.0f ticks (100 MHz: %.1f us)" % (
        level, cin, cout, tile, ru, rs, e0.elapsed_time(e1) * 1e3, t.shape[0], nsub.mean(), life.mean(), life.mean() / 100.0))
    for name, v in (("prologue (tables, zero-fill, first loads)", t[:, 1] - t[:, 0]), ("issue (positions, addresses, next gathers)", t[:, 3]),
                    ("operand wait (PCS_TRACE=2 only)", opw), ("MFMAs (+ operand waits unless PCS_TRACE=2)", t[:, 4]), ("ticket wait", tick),
                    ("commit", t[:, 6]), ("loop other", (t[:, 2] - t[:, 1]) - t[:, 3] - t[:, 4] - tick - t[:, 6] - opw),
                    ("final barrier + write-back", t_exit - t[:, 2])):
        print("   %-46s %5.1f %%   per sub-group %.0f ticks" % (name, 100.0 * v.sum() / tot, v.sum() / max(nsub.sum(), 1.0)))


if __name__ == "__main__":
    main()
