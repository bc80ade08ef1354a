"""Deterministic, name-keyed weights shared by tests/golden/make_golden.py and the tests."""
import zlib

import numpy as np
import torch


def seeded_state(model):
    """Deterministic weights that depend only on parameter NAMES and shapes (so that the test
    can regenerate them without storing 20+ MB)."""
    sd = model.state_dict()
    for name in sorted(sd):
        t = sd[name]
        if not t.dtype.is_floating_point:
            continue
        g = torch.Generator().manual_seed(zlib.crc32(name.encode()) & 0x7FFFFFFF)
        r = torch.rand(t.shape, generator=g)
        if name.endswith("running_var"):
            v = 0.5 + r
        elif name.endswith("running_mean"):
            v = 0.2 * (r - 0.5)
        elif t.dim() == 4:  # Conv2d (out, in, kh, kw) of RPVNet's range branch: its un-normalised 1x1 shortcuts must not amplify
            fan = t.shape[1] * t.shape[2] * t.shape[3]
            v = (r - 0.5) * 2.0 / np.sqrt(fan) * 1.7
        elif t.dim() >= 2:
            fan = t.shape[-2] * (t.shape[0] if t.dim() == 3 else 1)
            v = (r - 0.5) * 2.0 / np.sqrt(fan) * 1.7
        elif name.endswith("weight"):
            v = 0.75 + 0.5 * r
        else:
            v = 0.2 * (r - 0.5)
        sd[name] = v.to(t.dtype)
    model.load_state_dict(sd)
