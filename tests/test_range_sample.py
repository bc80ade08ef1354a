"""`range_to_point` of RPVNet (R:pcseg/model/segmentor/fusion/rpvnet/rpvnet.py:31-51: a python loop over frames around
torch.nn.functional.grid_sample(mode='bilinear')) -> csrc/rangesample.hip (one launch each way; the backward atomic-free).

Fixture: tests/golden/range_sample_golden.npz, made by `make_golden.py range_sample` by RUNNING the reference's function on CPU
(forward + autograd backward), incl. coordinates outside [-1, 1] (zero padding), on pixel centres and on the image border.
CPU: the NumPy oracle against the fixture. `-m gpu`: the HIP kernels against the fixture and, at BASELINE config 5's sizes
(4 frames of ~97 k points, 64 x 2048 down to 4 x 128 images, 56 ... 448 channels), against the oracle + adjoint / determinism
properties; the glue re-binding inside the reference's RPVNet module."""
import os

import numpy as np
import pytest
import torch

from oracle import oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def g():
    return np.load(os.path.join(ROOT, "tests", "golden", "range_sample_golden.npz"))


@pytest.mark.parametrize("case", ["a", "b"])
def test_oracle_matches_the_reference_function(g, case):
    img, pxpy, gout = g[case + "_img"], g[case + "_pxpy"], g[case + "_gout"]
    out = orc.range_sample_fwd(img, pxpy)
    assert np.abs(out - g[case + "_out"]).max() <= 2e-6 * max(1.0, np.abs(g[case + "_out"]).max())
    gimg = orc.range_sample_bwd(gout, pxpy, img.shape)
    assert np.abs(gimg - g[case + "_gimg"]).max() <= 1e-5 * np.abs(g[case + "_gimg"]).max()
    # rows of a frame outside [0, B) sample nothing (the reference's masks never select them)
    bad = pxpy.copy()
    bad[:7, 0] = img.shape[0]
    assert (orc.range_sample_fwd(img, bad)[:7] == 0).all()


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["a", "b"])
def test_hip_matches_the_reference_function(hip, g, case):
    from openpcseg_amd import rangelib
    img, pxpy, gout = g[case + "_img"], g[case + "_pxpy"], g[case + "_gout"]
    ti = _t(img).requires_grad_(True)
    out = rangelib.range_to_point(ti, _t(pxpy), "bilinear")
    assert out.shape == gout.shape
    assert float((out.detach().cpu() - torch.from_numpy(g[case + "_out"])).abs().max()) <= 2e-6 * max(1.0, np.abs(g[case + "_out"]).max())
    out.backward(_t(gout))
    ref = torch.from_numpy(g[case + "_gimg"])
    assert float((ti.grad.cpu() - ref).abs().max()) <= 1e-5 * float(ref.abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize("b,c,h,w,n", [(4, 56, 64, 2048, 390000), (4, 448, 4, 128, 390000), (2, 224, 16, 512, 150000), (3, 20, 8, 32, 5000),
                                       (1, 168, 64, 2048, 97000)])
def test_hip_range_sample_at_config5_sizes(hip, b, c, h, w, n):
    from openpcseg_amd import rangelib
    rng = np.random.default_rng(b * 1000 + c)
    img = rng.normal(size=(b, c, h, w)).astype(np.float32)
    frames = np.sort(rng.integers(0, b, size=n)).astype(np.float32)
    pxpy = np.concatenate([frames[:, None], rng.uniform(-1.02, 1.02, size=(n, 2)).astype(np.float32)], 1)
    gout = rng.normal(size=(n, c)).astype(np.float32)
    ti, tp, tg = _t(img).requires_grad_(True), _t(pxpy), _t(gout)
    out = rangelib.range_to_point(ti, tp)
    sel = rng.choice(n, size=min(n, 20000), replace=False)
    ref = orc.range_sample_fwd(img, pxpy[sel])
    assert np.abs(out.detach().cpu().numpy()[sel] - ref).max() <= 2e-6 * max(1.0, np.abs(ref).max())
    out.backward(tg)
    g1 = ti.grad.clone()
    # adjoint identity <S img, g> = <img, S^T g> in float64, determinism of the segmented backward
    lhs = float((out.detach().double() * tg.double()).sum())
    rhs = float((ti.detach().double() * g1.double()).sum())
    assert abs(lhs - rhs) <= 1e-6 * max(abs(lhs), float((out.detach().double().abs() * tg.double().abs()).sum()))
    ti.grad = None
    rangelib.range_to_point(ti, tp).backward(tg)
    assert torch.equal(ti.grad, g1)
    if n <= 160000:
        ref_g = orc.range_sample_bwd(gout, pxpy, img.shape)
        assert np.abs(g1.cpu().numpy() - ref_g).max() <= 1e-5 * np.abs(ref_g).max()
    # against torch's own op on the device, frame by frame (what the reference runs)
    t2 = _t(img).requires_grad_(True)
    parts = [torch.nn.functional.grid_sample(t2[f:f + 1], tp[tp[:, 0] == f][:, 1:][None, None], mode="bilinear", align_corners=False)
             .squeeze(0).squeeze(1).t() for f in range(b)]
    o2 = torch.cat(parts, 0)
    assert float((o2 - out).abs().max()) <= 2e-6 * max(1.0, float(o2.abs().max()))
    o2.backward(tg)
    assert float((t2.grad - g1).abs().max()) <= 2e-5 * float(g1.abs().max())   # torch's atomics: its own summation order


@pytest.mark.gpu
def test_range_to_point_detects_what_the_reference_would_reorder_without_a_host_read(hip):
    """Frames out of order: the kernels' result (input order) is not the reference's (grouped by frame). The order flag is computed on
    the device and looked at later -- no `.item()` on the critical path -- and a violation raises at the next verification point."""
    from openpcseg_amd import rangelib
    rangelib.verify_pending(block=True)
    img = torch.randn(2, 8, 8, 16, device="cuda")
    good = torch.tensor([[0, 0.1, 0.2], [0, -0.3, 0.5], [1, 0.7, -0.2]], device="cuda")
    out = rangelib.range_to_point(img, good, "bilinear")
    rangelib.verify_pending(block=True)                                  # in order: nothing to report
    assert out.shape == (3, 8)
    pxpy = torch.tensor([[1, 0.1, 0.2], [0, -0.3, 0.5], [1, 0.7, -0.2]], device="cuda")   # frames not grouped in ascending order
    rangelib.range_to_point(img, pxpy, "bilinear")
    with pytest.raises(RuntimeError, match="out of order"):
        rangelib.verify_pending(block=True)
    rangelib.verify_pending(block=True)                                  # reported once
    assert rangelib.range_to_point(img, good, "nearest", fallback=lambda f, p, m: m) == "nearest"   # other unsupported inputs: fallback
    with pytest.raises(RuntimeError):
        rangelib.range_to_point(img, good, "nearest")
