#!/usr/bin/env python3
"""Host-side cost of one training step of a tools/modelbench.py record, on a tiny batch (GPU time negligible: the wall time per
step IS the host's enqueue time of the same launch sequence), plus a cProfile of the step.
    PCS_MB_FRAMES=1 PCS_MB_POINTS=3000 python tools/host_profile_model.py minkunet34:fuse:bf16 [top_n]"""
import cProfile
import io
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("PCS_MB_FRAMES", "1")
os.environ.setdefault("PCS_MB_POINTS", "3000")
import torch  # noqa: E402

import modelbench  # noqa: E402


def main():
    spec = sys.argv[1]
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    name, source, dtype = (spec.split(":") + ["f32"])[:3]
    step = modelbench.bench_one(name, source, dtype, torch.device("cuda:0"), want_step=True)
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    n = 20
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    print("%s: host-bound step %.2f ms (tiny batch, %d steps)" % (spec, (time.perf_counter() - t0) / n * 1e3, n))
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    pr.disable()
    out = io.StringIO()
    pstats.Stats(pr, stream=out).sort_stats("tottime").print_stats(top)
    print("\n".join(l for l in out.getvalue().splitlines() if l.strip())[:9000])


if __name__ == "__main__":
    main()
