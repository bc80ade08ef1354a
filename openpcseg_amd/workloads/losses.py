"""Training loss of the MinkUNet workload: cross-entropy (label smoothing) + Lovasz-softmax,
the reference default (R:pcseg/model/segmentor/voxel/minkunet/minkunet.py:336-356,
R:pcseg/loss/__init__.py:106-115, R:tools/utils/common/lovasz_losses.py:23-35,158-228).
Dense torch ops on (N, num_class) logits -- outside the sparse hot path, kept in the timed
step so that no work of the reference's training iteration is skipped.
"""
import os

import torch
import torch.nn.functional as F


def lovasz_grad(gt_sorted):
    gts = gt_sorted.sum()
    inter = gts - gt_sorted.cumsum(0)
    union = gts + (1.0 - gt_sorted).cumsum(0)
    jac = 1.0 - inter / union
    if gt_sorted.numel() > 1:
        jac[1:] = jac[1:] - jac[:-1].clone()
    return jac


def lovasz_softmax_per_class(probas, labels, ignore=None):
    """classes='present', per_image=False: the reference's loop, one class at a time (lovasz_losses.py:176-204)."""
    if ignore is not None:
        keep = labels != ignore
        probas, labels = probas[keep], labels[keep]
    if probas.numel() == 0:
        return probas.sum() * 0.0
    losses = []
    present = (torch.bincount(labels, minlength=probas.shape[1]) > 0).tolist()  # one host sync, not one per class
    for c in range(probas.shape[1]):
        if not present[c]:
            continue
        fg = (labels == c).float()
        err = (fg - probas[:, c]).abs()
        err_sorted, perm = torch.sort(err, 0, descending=True)
        losses.append(torch.dot(err_sorted, lovasz_grad(fg[perm])))
    return torch.stack(losses).mean()


def lovasz_softmax(probas, labels, ignore=None):
    """The same function with the present classes as the rows of (C', n) tensors: one sort, one gather and a dozen
    elementwise kernels for all classes instead of ~25 small launches per class (19 classes x 1.4 M points: the
    per-class loop spent ~5 ms of a 70 ms bf16 step in launch-bound kernels). The Lovasz gradient depends on the
    labels only, so it is built outside autograd; (1 - fg).cumsum is position - fg.cumsum (integers, exact in fp32).
    Ignored points are not compacted away (a nonzero + gather forward, an index_put backward over (n, C)): their error
    is set to zero, which sorts them behind every point that matters -- each term they could touch is multiplied by
    a zero error, so the value and the gradient are the compacted ones."""
    if probas.numel() == 0:
        return probas.sum() * 0.0
    n, nc = probas.shape
    valid = None
    if ignore is None:
        # labels outside [0, nc) are invalid points here as well (what the HIP kernel does): neither counted nor given an error term
        valid = (labels >= 0) & (labels < nc)
        cnt = torch.bincount(torch.where(valid, labels, torch.full_like(labels, nc)), minlength=nc + 1)[:nc]
    else:
        # presence counts over the VALID points only: an out-of-range ignore label (255, -100, -1) must not be counted
        # into a clamped class (it would make that class "present" with no foreground point and add a max(err) term)
        # (integer counting into an extra bin: a float-weighted bincount is a non-deterministic CUDA op). Labels outside
        # [0, nc) that are NOT the ignore label (a negative label, 255 with ignore = 0) are invalid too: neither counted
        # nor allowed an error term.
        valid = (labels != ignore) & (labels >= 0) & (labels < nc)
        cnt = torch.bincount(torch.where(valid, labels, torch.full_like(labels, nc)), minlength=nc + 1)[:nc]
    cls = (cnt > 0).nonzero().squeeze(1)  # one host sync
    if cls.numel() == 0:
        return probas.sum() * 0.0
    pt = probas.t().index_select(0, cls)  # (C', n), rows = present classes
    with torch.no_grad():
        fg = (labels.unsqueeze(0) == cls.unsqueeze(1)).to(probas.dtype)
    err = (fg - pt).abs()
    if valid is not None:
        err = err * valid.to(probas.dtype).unsqueeze(0)
    err_sorted, perm = torch.sort(err, dim=1, descending=True)
    with torch.no_grad():
        fg_sorted = torch.gather(fg, 1, perm)
        gts = fg_sorted.sum(1, keepdim=True)
        cs = torch.empty_like(fg_sorted)
        for i in range(cs.shape[0]):  # a 2-D cumsum along the long axis is 15x slower than C' 1-D ones here
            torch.cumsum(fg_sorted[i], 0, out=cs[i])
        pos = torch.arange(1, n + 1, device=probas.device, dtype=probas.dtype)
        jac = 1.0 - (gts - cs) / (gts + (pos - cs))
        grad = jac.clone()
        if n > 1:
            grad[:, 1:] -= jac[:, :-1]
    return (err_sorted * grad).sum(1).mean()


class _LovaszSoftmaxHip(torch.autograd.Function):
    """The whole function in csrc/lovasz.hip (`pcs_lovasz_softmax_f32`): every class in one radix sort, the Jaccard
    gradient and d loss / d probas from the sorted stream; ~14 launches instead of ~320 through the torch form above
    and its autograd graph. Same tie order as the stable descending torch.sort of the batched form."""

    @staticmethod
    def forward(ctx, probas, labels, ignore):
        from .. import native
        loss, grad = native.backend().lovasz_softmax(probas.detach().float().contiguous(), labels.contiguous(), ignore,
                                                     need_grad=ctx.needs_input_grad[0])
        ctx.for_backwards = (grad, probas.dtype)
        return loss

    @staticmethod
    def backward(ctx, grad_out):
        grad, dtype = ctx.for_backwards
        return (grad * grad_out).to(dtype), None, None


def lovasz_softmax_device(probas, labels, ignore=None):
    """Device tensors the kernel serves ((n, C <= 60) probabilities): the fused HIP form; anything else -- host tensors (the
    oracle-backed CPU tests, the explicit CPU path), more than 60 classes -- the torch form of the same function."""
    if (probas.is_cuda and probas.dim() == 2 and probas.shape[1] <= 60 and labels.dim() == 1 and
            os.environ.get("PCS_LOVASZ_TORCH", "0") != "1"):
        return _LovaszSoftmaxHip.apply(probas, labels, ignore)
    return lovasz_softmax(probas, labels, ignore)


class SegLoss(torch.nn.Module):
    """CrossEntropyLoss(ignore_index, label_smoothing) + Lovasz-softmax on one shared log-softmax. The CE terms are
    written out (picked log-probability + smoothing term, masked mean) because torch's nll_loss reduction runs on a
    single workgroup (1.3 ms forward + 1.3 ms backward for 1.2 M rows); same value as torch.nn.CrossEntropyLoss."""

    def __init__(self, ignore_index=0, label_smoothing=0.0):
        super().__init__()
        self.ignore_index = ignore_index
        self.label_smoothing = float(label_smoothing)

    def forward(self, logits, target):
        logp = F.log_softmax(logits, dim=1)
        keep = target != self.ignore_index
        picked = logp.gather(1, target.clamp(0, logits.shape[1] - 1).unsqueeze(1)).squeeze(1)
        per_row = -(1.0 - self.label_smoothing) * picked
        if self.label_smoothing > 0:
            per_row = per_row - self.label_smoothing * logp.mean(dim=1)
        ce = (per_row * keep).sum() / keep.sum()
        lov = lovasz_softmax_per_class if os.environ.get("PCS_LOVASZ_LOOP", "0") == "1" else lovasz_softmax_device  # A/B
        return ce + lov(logp.exp(), target, ignore=self.ignore_index)
