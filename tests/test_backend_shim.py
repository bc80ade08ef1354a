"""Boundary B-B: the reference's OWN torchsparse/nn/functional/{hash,query,count,voxelize,devoxelize,conv}.py,
executed UNMODIFIED on top of this package's `torchsparse.backend` (all 20 names of TS:torchsparse/backend/pybind_cuda.cpp:18-39),
for host tensors (the `*_cpu` half, pure PyTorch) and for device tensors (the `*_cuda` half, C ABI). Their results are compared
with this package's operator API on the same inputs and with the reference goldens."""
import os
import sys
import types

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from stage_reference import TS_FUNCTIONAL, functional_source  # noqa: E402

NAMES_20 = [f + s for f in ("convolution_forward", "convolution_backward", "voxelize_forward", "voxelize_backward", "devoxelize_forward",
                            "devoxelize_backward", "hash", "kernel_hash", "hash_query", "count") for s in ("_cpu", "_cuda")]


@pytest.fixture()
def ref_functional():
    """The reference's functional modules, compiled from their unmodified source, importing `torchsparse` = this package's alias."""
    import openpcseg_amd
    had = {k: v for k, v in sys.modules.items() if k == "torchsparse" or k.startswith("torchsparse.")}
    openpcseg_amd.install_as_torchsparse()
    mods = {}
    for name in TS_FUNCTIONAL:
        src = functional_source(name)
        if src is None:
            pytest.skip("the reference's functional sources are not available (no /root/reference, nothing staged)")
        m = types.ModuleType("ref_ts_functional_" + name)
        exec(compile(src, "<TS:torchsparse/nn/functional/%s.py>" % name, "exec"), m.__dict__)
        mods[name] = m
    yield types.SimpleNamespace(**mods)
    for k in [k for k in sys.modules if k == "torchsparse" or k.startswith("torchsparse.")]:
        del sys.modules[k]
    sys.modules.update(had)


def test_backend_module_exports_the_twenty_names():
    import openpcseg_amd
    ts = openpcseg_amd.install_as_torchsparse()
    try:
        assert sorted(n for n in dir(ts.backend) if n.endswith(("_cpu", "_cuda"))) == sorted(NAMES_20)
    finally:
        for k in [k for k in sys.modules if k == "torchsparse" or k.startswith("torchsparse.")]:
            del sys.modules[k]


def _scene(golden, dev):
    coords = torch.from_numpy(golden["scene_coords"]).to(dev)
    return coords


def _run_all(ref, golden, dev, ours_ctx):
    """The six reference files on `dev` against this package's operators (under `ours_ctx`) and the goldens."""
    from openpcseg_amd import functional as F
    from openpcseg_amd.sparse import get_kernel_offsets
    coords = _scene(golden, dev)
    off = get_kernel_offsets(3, 1, 1, device=dev)
    # hash / kernel hash / query / count
    h = ref.hash.sphash(coords)
    hk = ref.hash.sphash(coords, off)
    q = torch.from_numpy(golden["query_q"]).to(dev)
    r = ref.query.sphashquery(q, h)
    assert np.array_equal(r.cpu().numpy(), golden["query_out"])
    with ours_ctx():
        assert torch.equal(h, F.sphash(coords)) and torch.equal(hk, F.sphash(coords, off))
        assert torch.equal(ref.query.sphashquery(hk, h), F.sphashquery(hk, h))
        idx = torch.randint(-1, 50, (4000,), generator=torch.Generator().manual_seed(1)).int().to(dev)
        assert torch.equal(ref.count.spcount(idx, 50), F.spcount(idx, 50))
        # voxelize / devoxelize, forward + backward
        n, m, c = 4000, 50, 16
        g = torch.Generator().manual_seed(2)
        feats = torch.randn(n, c, generator=g).to(dev)
        cnt = F.spcount(idx, m)
        for fn_ref, fn_our in ((ref.voxelize.spvoxelize, F.spvoxelize),):
            a, b = feats.clone().requires_grad_(True), feats.clone().requires_grad_(True)
            ya, yb = fn_ref(a, idx, cnt), fn_our(b, idx, cnt)
            gy = torch.randn(ya.shape, generator=g).to(dev)
            ya.backward(gy)
            yb.backward(gy)
            assert torch.allclose(ya, yb, rtol=1e-5, atol=1e-6) and torch.allclose(a.grad, b.grad, rtol=1e-5, atol=1e-6)
        vox = torch.randn(m, c, generator=g).to(dev)
        idx8 = torch.randint(-1, m, (n, 8), generator=g).int().to(dev)
        w8 = torch.rand(n, 8, generator=g).to(dev)
        a, b = vox.clone().requires_grad_(True), vox.clone().requires_grad_(True)
        ya, yb = ref.devoxelize.spdevoxelize(a, idx8, w8), F.spdevoxelize(b, idx8, w8)
        gy = torch.randn(ya.shape, generator=g).to(dev)
        ya.backward(gy)
        yb.backward(gy)
        assert torch.allclose(ya, yb, rtol=1e-5, atol=1e-5) and torch.allclose(a.grad, b.grad, rtol=1e-4, atol=1e-5)
        pts = torch.rand(n, 3, generator=g).to(dev) * 20
        tw = ref.devoxelize.calc_ti_weights(pts, idx8.t().contiguous(), scale=2)
        assert torch.allclose(tw, F.calc_ti_weights(pts, idx8.t().contiguous(), scale=2), rtol=1e-5, atol=1e-6)
    # the convolution Function of the reference's conv.py on the reference's goldens (forward, transposed, backward)
    for tag, name, transposed in [("conv_k3s1_N", "k3s1", False), ("conv_k2s2_N", "k2s2", False), ("conv_k2s2_T", "k2s2", True)]:
        nbmaps = torch.from_numpy(golden["kmap_%s_nbmaps" % name]).to(dev)
        nbsizes = torch.from_numpy(golden["kmap_%s_nbsizes" % name]).to(dev)
        x, w, gy = (torch.from_numpy(golden[tag + s]).to(dev) for s in ("_x", "_w", "_gy"))
        x, w = x.requires_grad_(True), w.requires_grad_(True)
        sizes = (golden[tag + "_y"].shape[0], golden[tag + "_y"].shape[0]) if transposed else (x.shape[0], golden[tag + "_y"].shape[0])
        y = ref.conv.ConvolutionFunction.apply(x, w, nbmaps, nbsizes, sizes, transposed)
        assert np.abs(y.detach().cpu().numpy() - golden[tag + "_y"]).max() <= 2e-5 * np.abs(golden[tag + "_y"]).max()
        y.backward(gy)
        assert np.abs(x.grad.cpu().numpy() - golden[tag + "_gx"]).max() <= 2e-5 * np.abs(golden[tag + "_gx"]).max()
        assert np.abs(w.grad.cpu().numpy() - golden[tag + "_gw"]).max() <= 2e-5 * np.abs(golden[tag + "_gw"]).max()


def test_reference_functional_files_run_on_the_cpu_half(ref_functional, golden):
    from openpcseg_amd import cpu_fallback
    _run_all(ref_functional, golden, "cpu", cpu_fallback.enabled)


@pytest.mark.gpu
def test_reference_functional_files_run_on_the_cuda_half(ref_functional, golden, hip):
    import contextlib
    _run_all(ref_functional, golden, "cuda", contextlib.nullcontext)
