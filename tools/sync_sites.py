#!/usr/bin/env python3
"""Where one training step of the bench workload makes the host wait for the device: torch's sync debug mode (item / tolist /
D2H copies / nonzero ...) plus torch.cuda.Event.synchronize and torch.cuda.synchronize, each with the innermost frames of this
repository that led there. Usage: python tools/sync_sites.py [--amp bf16]"""
import collections
import os
import sys
import traceback
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from openpcseg_amd import functional as pcsF  # noqa: E402,F401
from openpcseg_amd.sparse import SparseTensor  # noqa: E402
from openpcseg_amd.workloads.minkunet import MK34_LAYERS, MinkUNet  # noqa: E402
from openpcseg_amd.workloads.synthetic import make_batch  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def site():
    frames = [f for f in traceback.extract_stack()[:-2] if f.filename.startswith(ROOT) and "sync_sites" not in f.filename]
    return " <- ".join("%s:%d %s" % (os.path.relpath(f.filename, ROOT), f.lineno, f.name) for f in reversed(frames[-3:]))


def main():
    amp = torch.bfloat16 if "--amp" in sys.argv else None
    dev = torch.device("cuda:0")
    b = make_batch(list(range(12)))
    batch = {k: (v.to(dev) if hasattr(v, "to") else v) for k, v in b.items()}
    torch.manual_seed(0)
    model = MinkUNet(num_class=20, num_layer=MK34_LAYERS, cr=1.0).to(dev).train()
    params = [p for p in model.parameters() if p.requires_grad]
    opt = torch.optim.SGD(params, lr=0.02, momentum=0.9, weight_decay=1e-4, nesterov=True)

    def fresh():
        return {"lidar": SparseTensor(batch["lidar"].F, batch["lidar"].C), "targets": SparseTensor(batch["targets"].F, batch["targets"].C)}

    def step():
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=amp, enabled=amp is not None):
            out = model(fresh())
        out["loss"].backward()
        torch.nn.utils.clip_grad_norm_(params, 10.0)
        opt.step()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    counts = collections.Counter()
    ev_sync = torch.cuda.Event.synchronize

    def logged_event_sync(self):
        counts["Event.synchronize @ " + site()] += 1
        return ev_sync(self)

    torch.cuda.Event.synchronize = logged_event_sync
    torch.cuda.set_sync_debug_mode("warn")
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        orig_show = warnings.showwarning
        step()
    torch.cuda.set_sync_debug_mode("default")
    torch.cuda.Event.synchronize = ev_sync
    for w in rec:
        if "synchroniz" in str(w.message):
            counts["torch sync (%s:%d)" % (os.path.relpath(w.filename, ROOT) if w.filename.startswith(ROOT) else w.filename, w.lineno)] += 1
    print("host waits in one %s step:" % ("bf16" if amp else "fp32"))
    for k, v in sorted(counts.items(), key=lambda kv: -kv[1]):
        print("  x%-3d %s" % (v, k))


if __name__ == "__main__":
    main()
