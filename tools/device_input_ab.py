#!/usr/bin/env python3
"""What the device input pipeline costs inside the bf16 training step of the headline workload: the pre-voxelised batch (base),
device_collate inside the step (round 5) and the side-stream prefetch (round 6), alternated so that all three see the same
clock state.  python tools/device_input_ab.py [steps per block, default 10] [blocks, default 3]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from openpcseg_amd.sparse import SparseTensor  # noqa: E402
from openpcseg_amd.workloads.minkunet import MK34_LAYERS, MinkUNet  # noqa: E402
from openpcseg_amd.workloads.synthetic import DeviceInputPrefetcher, device_collate, make_batch, make_raw_batch  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    dev = torch.device("cuda:0")
    seeds = list(range(12))
    b = make_batch(seeds)
    f, c, l = b["lidar"].F.to(dev), b["lidar"].C.to(dev), b["targets"].F.to(dev)
    raw = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in make_raw_batch(seeds).items()}
    torch.manual_seed(0)
    model = MinkUNet(num_class=20, num_layer=MK34_LAYERS, cr=1.0).to(dev).train()
    params = [p for p in model.parameters() if p.requires_grad]
    opt = torch.optim.SGD(params, lr=0.01, momentum=0.9, weight_decay=1e-4, nesterov=True)
    pipe = DeviceInputPrefetcher(lambda i: raw)
    mode = {"m": "base"}

    def step():
        opt.zero_grad(set_to_none=True)
        if mode["m"] == "base":
            batch = {"lidar": SparseTensor(f.view_as(f), c.view_as(c)), "targets": SparseTensor(l.view_as(l), c.view_as(c))}
        elif mode["m"] == "inline":
            batch = device_collate(raw)
        else:
            batch = pipe.next()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = model(batch)
        out["loss"].backward()
        torch.nn.utils.clip_grad_norm_(params, 10.0)
        opt.step()
        if mode["m"] == "prefetch":
            pipe.prefetch()

    t_end = time.perf_counter() + 40.0
    while time.perf_counter() < t_end:   # clock pre-heat
        step()
    torch.cuda.synchronize()
    res = {"base": [], "inline": [], "prefetch": []}
    for _ in range(blocks):
        for m in ("base", "inline", "prefetch"):
            mode["m"] = m
            step(); step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
            torch.cuda.synchronize()
            res[m].append((time.perf_counter() - t0) / steps * 1e3)
    for m, v in res.items():
        print("%-9s ms/step: %s  (median %.2f)" % (m, " ".join("%.2f" % x for x in v), sorted(v)[len(v) // 2]))
    med = {m: sorted(v)[len(v) // 2] for m, v in res.items()}
    print("input cost: inline %.2f ms, prefetch %.2f ms per step" % (med["inline"] - med["base"], med["prefetch"] - med["base"]))


if __name__ == "__main__":
    main()
