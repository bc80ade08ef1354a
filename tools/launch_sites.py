#!/usr/bin/env python3
"""Where the small launches of a training step come from: torch.profiler with python stacks over a few bench steps, the
runtime memcpy / memset calls and the copy_ / zero_ / fill_ / item ops grouped by their innermost frames inside this repo.
Usage: python tools/launch_sites.py [off|bf16] [steps]  -> table on stdout."""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

from bench import fresh, to_device  # noqa: E402
from openpcseg_amd.workloads.minkunet import MK34_LAYERS, MinkUNet  # noqa: E402
from openpcseg_amd.workloads.synthetic import make_batch  # noqa: E402

WATCH = ("aten::copy_", "aten::zero_", "aten::fill_", "aten::item", "aten::_local_scalar_dense", "hipMemcpyAsync", "hipMemsetAsync",
         "hipMemcpyWithStream", "hipMemcpy", "hipMemset", "hipMemsetD8Async", "hipMemsetD32Async", "aten::zeros", "aten::clone", "aten::contiguous")


def main():
    amp = {"bf16": torch.bfloat16}.get(sys.argv[1]) if len(sys.argv) > 1 else None
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = MinkUNet(num_class=20, num_layer=MK34_LAYERS, cr=1.0).to(dev).train()
    opt = torch.optim.SGD(model.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4, nesterov=True)
    batch = to_device(make_batch(list(range(12))), dev)

    def step():
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=amp, enabled=amp is not None):
            out = model(fresh(batch))
        out["loss"].backward()
        opt.step()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    agg = collections.Counter()
    names = collections.Counter()
    for ev in prof.events():
        names[ev.name] += 1
        if ev.name not in WATCH:
            continue
        frames = [f for f in (ev.stack or []) if root in f or "openpcseg_amd" in f or "bench.py" in f]
        site = " <- ".join(f.replace(root + "/", "") for f in frames[:3])
        if not site:   # no python stacks on this build: the chain of enclosing CPU ops (autograd Function names included)
            chain, par = [], getattr(ev, "cpu_parent", None)
            while par is not None and len(chain) < 6:
                chain.append(par.name[:48])
                par = getattr(par, "cpu_parent", None)
            site = " <- ".join(chain) or "(top level)"
        agg[(ev.name, site)] += 1
    print("events per step by name (top 40):")
    for n, c in names.most_common(40):
        print("  %8.1f  %s" % (c / steps, n[:110]))
    print("\nwatched ops per step by call site:")
    for (n, site), c in agg.most_common(80):
        print("  %7.1f  %-28s %s" % (c / steps, n, site[:260]))


if __name__ == "__main__":
    main()
