#!/usr/bin/env python3
"""Which torch ops (outside the C-ABI library) cost device time in one training step: torch.profiler, self device time
per (aten op, input shapes). Usage: python tools/step_op_profile.py [bf16|fp16]"""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

from openpcseg_amd.sparse import SparseTensor  # noqa: E402
from openpcseg_amd.workloads.minkunet import MK34_LAYERS, MinkUNet  # noqa: E402
from openpcseg_amd.workloads.synthetic import make_batch  # noqa: E402


def main():
    amp = {"bf16": torch.bfloat16, "fp16": torch.float16}.get(sys.argv[1]) if len(sys.argv) > 1 else None
    frames = int(os.environ.get("PCS_PROFILE_FRAMES", "12"))
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = MinkUNet(num_class=20, num_layer=MK34_LAYERS, cr=1.0).to(dev).train()
    opt = torch.optim.SGD(model.parameters(), lr=0.01, momentum=0.9)
    b = make_batch(list(range(frames)))
    coords = b["lidar"].C.to(dev)
    feats, tg = b["lidar"].F.to(dev), b["targets"].F.to(dev)

    def step():
        batch = {"lidar": SparseTensor(feats, coords), "targets": SparseTensor(tg, coords)}
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=amp, enabled=amp is not None):
            out = model(batch)
        out["loss"].backward()
        opt.step()

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        step()
        torch.cuda.synchronize()
    agg = collections.defaultdict(lambda: [0.0, 0])
    for ev in prof.key_averages(group_by_input_shape=True):
        dt = getattr(ev, "self_device_time_total", None)
        if dt is None:
            dt = getattr(ev, "self_cuda_time_total", 0)
        if not dt or not ev.key.startswith("aten::"):
            continue
        a = agg[(ev.key, str(ev.input_shapes)[:110])]
        a[0] += dt
        a[1] += ev.count
    rows = sorted(agg.items(), key=lambda kv: -kv[1][0])
    tot = sum(v[0] for _, v in rows)
    print("aten ops, self device time: %.2f ms in one step" % (tot / 1e3))
    for (name, shp), (us, n) in rows[:70]:
        print("%8.3f ms %5d  %-24s %s" % (us / 1e3, n, name, shp))


if __name__ == "__main__":
    main()
