#!/usr/bin/env python3
"""A/B of the half-precision fused convolution: the wave-autonomous kernel (conv_wave5h.hip, PCS_CONVH_RING=0) against the
column-parallel ring kernel (conv_ring6h.hip), on the real rulebooks of the synthetic batch.

  python tools/ring_ab.py                   -> runs itself once per mode (the switch is read once per process) and prints a table
  python tools/ring_ab.py --one             -> one mode (the environment as it is), one line per shape

Every line also carries the relative difference to the fp32 MFMA convolution of the same inputs (parity at a glance).
PCS_AB_FRAMES (12), PCS_AB_REPS (30), PCS_AB_SHAPES ("level cin cout;...")."""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

DEFAULT_SHAPES = "0 96 96;0 128 96;1 96 96;1 128 96;2 128 128;2 192 128;3 256 256;3 384 256;3 128 128;4 256 256;2 64 128"


def one():
    import torch
    from openpcseg_amd import functional as F
    from openpcseg_amd import native
    from openpcseg_amd.workloads.synthetic import make_batch
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
    maps = {}
    for level, cin, cout in shapes:
        c = levels[level]
        if level not in maps:
            maps[level] = F.build_kernel_map(c, c, (3, 3, 3), (2 ** level,) * 3, (1, 1, 1))
        entry = maps[level]
        n, p = c.shape[0], entry.fwd.num_pairs
        torch.manual_seed(level * 1000 + cin + cout)
        x = torch.randn(n, cin, device=dev)
        w = torch.randn(27, cin, cout, device=dev) * 0.05
        xh = x.to(torch.bfloat16)
        wp = be.prepare_weights_h(w, torch.bfloat16, transpose=False)
        ref = be.conv_gather_gemm(xh.float(), w.to(torch.bfloat16).float(), entry.fwd)
        run = lambda: be.conv_gather_gemm_h(xh, wp, 27, cout, entry.fwd)
        y = run().float()
        err = float((y - ref).abs().max() / ref.abs().max())
        for _ in range(10):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        t = be.tile_rows(cin, cout, entry.fwd, 1)
        print("AB level=%d n=%d pairs=%d %d->%d tile=%d us=%.1f tflops=%.1f err=%.2e" %
              (level, n, p, cin, cout, t, us, 2.0 * p * cin * cout / us / 1e6, err), flush=True)


def main():
    if "--one" in sys.argv:
        return one()
    modes = [("wave5h", {"PCS_CONVH_RING": "0"}), ("ring nc2", {"PCS_CONVH_RING": "1", "PCS_CONVH_RING_NC": "2"}),
             ("ring nc1", {"PCS_CONVH_RING": "1", "PCS_CONVH_RING_NC": "1"})]
    rows = {}
    for name, env in modes:
        e = dict(os.environ)
        e.update(env)
        try:
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--one"], env=e, capture_output=True, text=True,
                                 timeout=int(os.environ.get("PCS_AB_TIMEOUT", "240")))
            text = out.stdout
            if out.returncode != 0:
                print("mode %s failed (rc %d): %s" % (name, out.returncode, out.stderr[-800:]))
        except subprocess.TimeoutExpired as ex:
            text = (ex.stdout or b"").decode() if isinstance(ex.stdout, bytes) else (ex.stdout or "")
            print("mode %s TIMED OUT" % name)
        for line in text.splitlines():
            if line.startswith("AB "):
                kv = dict(f.split("=") for f in line.split()[1:] if "=" in f)
                key = (kv["level"], line.split()[4])
                rows.setdefault(key, {})[name] = kv
    names = [m[0] for m in modes]
    print("| level | cin->cout | " + " | ".join("%s us (tile) TFLOP/s err" % n for n in names) + " |")
    print("|---|---|" + "---|" * len(names))
    for (level, shape), r in rows.items():
        cells = []
        for n in names:
            kv = r.get(n)
            cells.append("%s (%s) %s %s" % (kv["us"], kv["tile"], kv["tflops"], kv["err"]) if kv else "-")
        print("| %s | %s | %s |" % (level, shape, " | ".join(cells)))


if __name__ == "__main__":
    main()
