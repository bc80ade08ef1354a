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


def lovasz_grad_mismatch(probas, labels, ignore, ga, gb):
    """Largest difference between two Lovasz-softmax gradients (n, C) where the function defines them: element by element
    where a point's error is unique within its class, and as the SUM of the Jaccard weights (the gradient with the sign
    of d|fg - p| / dp taken off) over each group of equal errors otherwise: the order inside a tie group -- saturated
    probabilities -- is the sort implementation's, the group's total weight is not."""
    import numpy as np
    p = np.asarray(probas, dtype=np.float32)
    lab = np.asarray(labels).astype(np.int64)
    n, nc = p.shape
    valid = (lab >= 0) & (lab < nc)
    if ignore is not None:
        valid &= lab != ignore
    worst = float(np.abs(np.asarray(ga)[~valid] - np.asarray(gb)[~valid]).max()) if (~valid).any() else 0.0
    rows = np.nonzero(valid)[0]
    for c in range(nc):
        fg = lab[rows] == c
        err = np.abs(fg.astype(np.float32) - p[rows, c])
        sgn = np.where(fg, -1.0, 1.0)
        _, inv = np.unique(err, return_inverse=True)
        sa = np.bincount(inv, weights=sgn * np.asarray(ga, dtype=np.float64)[rows, c])
        sb = np.bincount(inv, weights=sgn * np.asarray(gb, dtype=np.float64)[rows, c])
        if len(sa):
            worst = max(worst, float(np.abs(sa - sb).max()))
    return worst
