"""Training loss of the MinkUNet workload: cross-entropy (label smoothing) + Lovasz-softmax,
the reference default (R:pcseg/model/segmentor/voxel/minkunet/minkunet.py:336-356,
R:pcseg/loss/__init__.py:106-115, R:tools/utils/common/lovasz_losses.py:23-35,158-228).
Dense torch ops on (N, num_class) logits -- outside the sparse hot path, kept in the timed
step so that no work of the reference's training iteration is skipped.
"""
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


def lovasz_softmax(probas, labels, ignore=None):
    """classes='present', per_image=False."""
    if ignore is not None:
        keep = labels != ignore
        probas, labels = probas[keep], labels[keep]
    if probas.numel() == 0:
        return probas.sum() * 0.0
    losses = []
    for c in range(probas.shape[1]):
        fg = (labels == c).float()
        if fg.sum() == 0:
            continue
        err = (fg - probas[:, c]).abs()
        err_sorted, perm = torch.sort(err, 0, descending=True)
        losses.append(torch.dot(err_sorted, lovasz_grad(fg[perm])))
    return torch.stack(losses).mean()


class SegLoss(torch.nn.Module):
    def __init__(self, ignore_index=0, label_smoothing=0.0):
        super().__init__()
        self.ignore_index = ignore_index
        self.ce = torch.nn.CrossEntropyLoss(ignore_index=ignore_index, label_smoothing=label_smoothing)

    def forward(self, logits, target):
        return self.ce(logits, target) + lovasz_softmax(logits.softmax(dim=1), target, ignore=self.ignore_index)
