#!/usr/bin/env python3
"""Per-module outputs of a reference segmentor (BASELINE configs 2-5, tests/golden/fullsize.py), to localise a parity gap.
  python tools/module_trace.py ref <config> <rays> <out.npz>    (build container: the reference on its own CPU backend)
  python tools/module_trace.py hip <config> <rays> <ref.npz>    (GPU box: the same model on libpcseg_hip.so, compared module by module)
Outputs are kept for every module up to depth 2, every 64th row."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import fullsize as fs  # noqa: E402
import make_golden as mg  # noqa: E402
from seeded import seeded_state  # noqa: E402


def main():
    mode, cfgn, n, path = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    n = None if n <= 0 else n
    if mode == "ref":
        ts = mg.import_reference_torchsparse()
        mg.install_scatter_stub()
        rnf = mg.install_range_stub()
        ST, dev = ts.SparseTensor, torch.device("cpu")
        torch.Tensor.cuda = lambda self, *a, **k: self
    else:
        import openpcseg_amd
        openpcseg_amd.install_reference_aliases()
        from openpcseg_amd.sparse import SparseTensor as ST
        dev = torch.device("cuda:0")
    dotted, cls = fs.MODEL_PATH[cfgn]
    mod = mg.import_reference_model(dotted)
    if cfgn == "config5":
        mod.rnf = rnf if mode == "ref" else sys.modules["range_utils.nn.functional"]
    if mode != "ref":
        for m in list(sys.modules.values()):
            if getattr(m, "__name__", "").startswith(("pcseg.", "tools.")) and hasattr(m, "torch_scatter"):
                m.torch_scatter = sys.modules["torch_scatter"]
    torch.manual_seed(0)
    model = getattr(mod, cls)(mg._AttrDict(fs.MODEL_CFG[cfgn]), 20)
    seeded_state(model)
    model.to(dev).train()
    fs.freeze_dropout(model)
    batch = fs.to_device(cfgn, fs.build_inputs(cfgn, ST, n), dev, ST)
    stats, order = {}, []

    def hook(name):
        def f(m, i, o):
            x = o.F if hasattr(o, "F") else (o[0] if isinstance(o, tuple) else o)
            if hasattr(x, "F"):
                x = x.F
            if isinstance(x, torch.Tensor) and x.is_floating_point() and x.dim() >= 1:
                key = "%s#%d" % (name, sum(1 for k in order if k.startswith(name + "#")))
                order.append(key)
                stats[key] = x.detach().float().cpu().numpy()[::64].copy()
        return f
    for name, m in model.named_modules():
        if name and name.count(".") <= 1:
            m.register_forward_hook(hook(name))
    with torch.no_grad():
        model(batch)
    if mode == "ref":
        np.savez_compressed(path, _order=np.array(order), **stats)
        print("wrote", path, len(order), "module outputs")
        return
    ref = np.load(path)
    print("%-44s %10s %12s %10s" % ("module", "scale", "max abs diff", "relative"))
    for key in [str(k) for k in ref["_order"]]:
        if key not in stats:
            print("%-44s missing on this side" % key)
            continue
        a, b = ref[key], stats[key]
        if a.shape != b.shape:
            print("%-44s shape %s vs %s" % (key, a.shape, b.shape))
            continue
        sc = float(np.abs(a).max())
        d = float(np.abs(a.astype(np.float64) - b).max())
        print("%-44s %10.3g %12.3g %10.2g" % (key, sc, d, d / max(sc, 1e-30)))


if __name__ == "__main__":
    main()
