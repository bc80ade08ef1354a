#!/usr/bin/env python3
"""Lovasz-softmax of the criterion, forward + backward to the logits, on a bench-sized batch: csrc/lovasz.hip (one sort for all
classes) beside the batched torch form and the reference's per-class loop. Usage: python tools/lovasz_bench.py [points] [classes]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from openpcseg_amd.workloads.losses import lovasz_softmax, lovasz_softmax_device, lovasz_softmax_per_class  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1160000
    nc = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    logits = (torch.randn(n, nc, device=dev) * 3).requires_grad_(True)
    target = torch.randint(0, nc, (n,), device=dev)
    for name, fn in (("hip one-sort", lovasz_softmax_device), ("torch batched", lovasz_softmax), ("torch per-class loop", lovasz_softmax_per_class)):
        def step():
            logits.grad = None
            loss = fn(torch.log_softmax(logits, 1).exp(), target, ignore=0)
            loss.backward()
            return loss
        for _ in range(3):
            val = step()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(10):
            step()
        e1.record()
        torch.cuda.synchronize()
        print("%-22s %8.3f ms / step (softmax + Lovasz forward + backward), loss %.7f, |grad|_1 %.7e"
              % (name, e0.elapsed_time(e1) / 10, float(val), float(logits.grad.abs().sum())))


if __name__ == "__main__":
    main()
