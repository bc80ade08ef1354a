#!/usr/bin/env python3
"""Host-side enqueue time of one training step vs its GPU time (is the step launch-bound?).
Usage: python tools/host_overhead.py [steps] [frames] [points_per_frame]   (a tiny batch, e.g. 1 frame x 3000 points,
makes the GPU work negligible: the wall time per step is then the pure host cost of the same launch sequence)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bench import fresh, to_device  # noqa: E402
from openpcseg_amd.workloads.minkunet import MK34_LAYERS, MinkUNet  # noqa: E402
from openpcseg_amd.workloads.synthetic import make_batch  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    npts = int(sys.argv[3]) if len(sys.argv) > 3 else None
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = MinkUNet(num_class=20, num_layer=MK34_LAYERS).to(dev).train()
    params = [p for p in model.parameters() if p.requires_grad]
    opt = torch.optim.SGD(params, lr=0.03, momentum=0.9, weight_decay=1e-4, nesterov=True)
    batch = to_device(make_batch(list(range(frames)), n_points=npts) if npts else make_batch(list(range(frames))), dev)

    def step():
        opt.zero_grad(set_to_none=True)
        out = model(fresh(batch))
        t1 = time.perf_counter()
        out["loss"].backward()
        t2 = time.perf_counter()
        torch.nn.utils.clip_grad_norm_(params, 10.0)
        opt.step()
        return t1, t2

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    enq = fwd = bwd = 0.0
    t_all = time.perf_counter()
    for _ in range(steps):
        t0 = time.perf_counter()
        t1, t2 = step()
        t3 = time.perf_counter()
        enq += t3 - t0; fwd += t1 - t0; bwd += t2 - t1
        torch.cuda.synchronize()  # drain, so that the next step's enqueue time is not hidden behind a full queue
    total = time.perf_counter() - t_all
    print("per step: host enqueue %.1f ms (forward incl. its host syncs %.1f, backward %.1f, clip+SGD %.1f); wall with a drain per step %.1f ms"
          % (1e3 * enq / steps, 1e3 * fwd / steps, 1e3 * bwd / steps, 1e3 * (enq - fwd - bwd) / steps, 1e3 * total / steps))


if __name__ == "__main__":
    main()
