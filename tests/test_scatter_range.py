"""Cylinder scatter (torch_scatter semantics, PARITY UNPINNED -- restated) and range_lib K13-K15.
CPU part: the restatement against closed-form expectations and the reference's only known-answer-ish
artefact for this path (RL:example.py:10-24, pxpy = [[0,2,2],[0,2,2],[1,1,0]]), plus the import aliases;
[r5] range_lib is now PINNED: oracle/build_ref_rangelib.py runs the reference's own CUDA kernel text on the host (launches rewritten
mechanically, a prelude supplies blockIdx / atomicAdd) -- the restatement and the HIP kernels are compared with THAT.
GPU part (-m gpu): HIP kernels vs the restatement and the host-run reference kernels, gradcheck-style adjoint identities."""
import numpy as np
import pytest
import torch

from oracle import oracle as orc


def test_rangelib_example_known_answer():
    pxpy = np.array([[0, 2, 2], [0, 2, 2], [1, 1, 0]], dtype=np.int32)  # (batch, px, py), RL:example.py:12
    cm = orc.map_count(pxpy, 2, 5, 4)
    assert cm.shape == (2, 5, 4) and cm.sum() == 3 and cm[0, 2, 2] == 2 and cm[1, 0, 1] == 1
    pf = np.array([[1, 1, 2], [1, 2, 2], [4, 4, 4]], dtype=np.float32)
    fm = orc.denselize_fwd(pf, cm, pxpy)
    assert fm.shape == (2, 3, 5, 4)
    assert np.allclose(fm[0, :, 2, 2], [1.0, 1.5, 2.0]) and np.allclose(fm[1, :, 0, 1], [4, 4, 4])
    g = np.random.default_rng(0).normal(size=fm.shape).astype(np.float32)
    # adjoint: <denselize(f), g> == <f, denselize_bwd(g)>   (what the reference's gradcheck verifies)
    assert np.isclose((fm * g).sum(), (pf * orc.denselize_bwd(g, cm, pxpy)).sum(), rtol=1e-5)


@pytest.fixture(scope="module")
def ref_rangelib():
    from oracle import build_ref_rangelib as rl
    if rl.load() is None:
        pytest.skip("oracle/_ref/ref_rangelib.so not built (needs /root/reference)")
    return rl


def _range_case(seed, n, b, c, h, w):
    """In-image coordinates incl. repeated pixels and empty pixels (the reference kernels have no upper-bound check,
    RL:range_utils/src/denselize_gpu.cu:17 'TODO': rows outside the image are undefined behaviour there, not a parity case)."""
    rng = np.random.default_rng(seed)
    pxpy = np.stack([rng.integers(0, b, n), rng.integers(0, w, n), rng.integers(0, h, n)], 1).astype(np.int32)
    pxpy[: n // 10, 1:] = pxpy[n // 10: 2 * (n // 10), 1:]     # more collisions
    return pxpy, rng.normal(size=(n, c)).astype(np.float32), rng.normal(size=(b, c, h, w)).astype(np.float32)


@pytest.mark.parametrize("seed,n,b,c,h,w", [(0, 5000, 3, 12, 16, 64), (1, 20000, 2, 32, 8, 128), (2, 300, 1, 5, 4, 16)])
def test_rangelib_oracle_matches_the_reference_kernels_run_on_the_host(ref_rangelib, seed, n, b, c, h, w):
    """K13-K15 pinned to the reference's own source: map_count bit-exact, denselize forward / backward within float rounding of the
    reference's serial atomicAdd order; and the example of RL:example.py through the reference's kernels."""
    pxpy, feat, g = _range_case(seed, n, b, c, h, w)
    cm = ref_rangelib.map_count(pxpy, b, h, w)
    assert np.array_equal(cm, orc.map_count(pxpy, b, h, w)) and int(cm.sum()) == n
    fm = ref_rangelib.denselize_fwd(feat, cm, pxpy)
    assert np.abs(fm - orc.denselize_fwd(feat, cm, pxpy)).max() <= 2e-6 * np.abs(fm).max()
    gb = ref_rangelib.denselize_bwd(g, cm, pxpy)
    assert np.abs(gb - orc.denselize_bwd(g, cm, pxpy)).max() <= 1e-6 * np.abs(gb).max()
    ex = np.array([[0, 2, 2], [0, 2, 2], [1, 1, 0]], dtype=np.int32)
    ecm = ref_rangelib.map_count(ex, 2, 5, 4)
    assert ecm[0, 2, 2] == 2 and ecm[1, 0, 1] == 1 and ecm.sum() == 3
    efm = ref_rangelib.denselize_fwd(np.array([[1, 1, 2], [1, 2, 2], [4, 4, 4]], np.float32), ecm, ex)
    assert np.allclose(efm[0, :, 2, 2], [1.0, 1.5, 2.0]) and np.allclose(efm[1, :, 0, 1], [4, 4, 4])


def test_scatter_max_restatement():
    rng = np.random.default_rng(1)
    src = rng.normal(size=(500, 7)).astype(np.float32)
    idx = rng.integers(0, 40, size=500)
    out, arg = orc.scatter_max(src, idx, 41)
    for v in range(41):
        rows = src[idx == v]
        if rows.size:
            assert np.array_equal(out[v], rows.max(axis=0))
            assert np.array_equal(src[arg[v], np.arange(7)], out[v])
        else:
            assert (out[v] == 0).all() and (arg[v] == -1).all()
    t = torch.from_numpy(src).requires_grad_(True)
    ref = torch.zeros(41, 7).scatter_reduce(0, torch.from_numpy(idx)[:, None].expand(-1, 7), t, "amax", include_self=False)
    assert np.allclose(ref.detach().numpy()[:40], out[:40])


def test_reference_aliases_resolve(oracle_backend):
    import openpcseg_amd
    openpcseg_amd.install_reference_aliases()
    import range_utils.nn.functional as rnf
    import torch_scatter
    import torchsparse.backend as tb
    src = torch.randn(50, 4, requires_grad=True)
    idx = torch.randint(0, 9, (50,))
    out, arg = torch_scatter.scatter_max(src, idx, dim=0)
    out.sum().backward()
    assert out.shape[1] == 4 and src.grad.sum().item() == pytest.approx(float((arg >= 0).sum()))
    mean = torch_scatter.scatter_mean(src.detach(), idx, dim=0)
    for v in range(mean.shape[0]):
        if (idx == v).any():
            assert torch.allclose(mean[v], src.detach()[idx == v].mean(0), atol=1e-6)
    pxpy = torch.tensor([[0, 2, 2], [0, 2, 2], [1, 1, 0]], dtype=torch.int32)
    cm = rnf.map_count(pxpy, 2, 5, 4)
    pf = torch.tensor([[1., 1, 2], [1, 2, 2], [4, 4, 4]], requires_grad=True)
    fm = rnf.denselize(pf, cm, pxpy)
    fm.sum().backward()
    assert torch.allclose(pf.grad, torch.tensor([[.5, .5, .5], [.5, .5, .5], [1, 1, 1]]))
    assert len([n for n in dir(tb) if n.endswith("_cuda")]) == 10


# ---- GPU ------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("c", [5, 32, 256])
def test_hip_scatter_max(hip, c):
    rng = np.random.default_rng(c)
    n, m = 40000, 6000
    src = rng.normal(size=(n, c)).astype(np.float32)
    idx = rng.integers(0, m - 3, size=n)  # last voxels stay empty
    out, arg = hip.scatter_max_fwd(torch.from_numpy(src).cuda(), torch.from_numpy(idx).cuda(), m)
    eo, ea = orc.scatter_max(src, idx, m)
    assert np.array_equal(out.cpu().numpy(), eo)
    a = arg.cpu().numpy()
    assert ((a >= 0) == (ea >= 0)).all()
    vv, jj = np.nonzero(a >= 0)
    assert np.array_equal(src[a[vv, jj], jj], eo[vv, jj])  # arg points at a maximal element (ties: any)
    g = rng.normal(size=(m, c)).astype(np.float32)
    gs = hip.scatter_max_bwd(torch.from_numpy(g).cuda(), arg, n).cpu().numpy()
    assert np.allclose(gs, orc.scatter_max_bwd(g, a, n))


@pytest.mark.gpu
@pytest.mark.parametrize("n,B,C,H,W", [(50000, 3, 24, 16, 128), (20000, 2, 80, 7, 100), (3000, 1, 5, 64, 512), (0, 1, 8, 4, 64)])
def test_hip_rangelib(hip, n, B, C, H, W):
    rng = np.random.default_rng(3)
    pxpy = np.stack([rng.integers(0, B, n), rng.integers(-2, W + 2, n), rng.integers(-1, H + 1, n)], 1).astype(np.int32)
    feat = rng.normal(size=(n, C)).astype(np.float32)
    tp, tf = torch.from_numpy(pxpy).cuda(), torch.from_numpy(feat).cuda()
    cm = hip.map_count(tp, B, H, W)
    ecm = orc.map_count(pxpy, B, H, W)
    assert np.array_equal(cm.cpu().numpy(), ecm)
    dfm = hip.denselize_fwd(tf, cm, tp)  # segmented over the points sorted by pixel; every element written once
    fm = dfm.cpu().numpy()
    efm = orc.denselize_fwd(feat, ecm, pxpy)
    assert np.abs(fm - efm).max() <= 1e-5 * max(np.abs(efm).max(), 1e-30)
    if C % 4 == 0:  # (odd channel counts take the atomic form)
        assert torch.equal(dfm, hip.denselize_fwd(tf, cm, tp))  # deterministic
    afm = hip.denselize_fwd_atomic(tf, cm, tp).cpu().numpy()  # the reference's atomic dataflow
    assert np.abs(afm - efm).max() <= 1e-5 * max(np.abs(efm).max(), 1e-30)
    g = rng.normal(size=efm.shape).astype(np.float32)
    egb = orc.denselize_bwd(g, ecm, pxpy)
    gb = hip.denselize_bwd(torch.from_numpy(g).cuda(), cm, tp).cpu().numpy()
    assert np.allclose(gb, egb, rtol=1e-6, atol=1e-7)
    assert np.allclose(hip.denselize_bwd_gather(torch.from_numpy(g).cuda(), cm, tp).cpu().numpy(), egb, rtol=1e-6, atol=1e-7)
    ex = np.array([[0, 2, 2], [0, 2, 2], [1, 1, 0]], dtype=np.int32)  # RL:example.py
    assert np.array_equal(hip.map_count(torch.from_numpy(ex).cuda(), 2, 5, 4).cpu().numpy(), orc.map_count(ex, 2, 5, 4))


@pytest.mark.gpu
@pytest.mark.parametrize("seed,n,b,c,h,w", [(0, 50000, 3, 24, 16, 128), (1, 20000, 2, 56, 64, 256), (2, 3000, 1, 5, 8, 64)])
def test_hip_rangelib_against_the_reference_kernels_run_on_the_host(hip, ref_rangelib, seed, n, b, c, h, w):
    """The HIP kernels directly against the reference's own kernel text executed on the host (oracle/_ref/ref_rangelib.so)."""
    pxpy, feat, g = _range_case(seed, n, b, c, h, w)
    tp = torch.from_numpy(pxpy).cuda()
    cm = hip.map_count(tp, b, h, w)
    rcm = ref_rangelib.map_count(pxpy, b, h, w)
    assert np.array_equal(cm.cpu().numpy(), rcm)
    rfm = ref_rangelib.denselize_fwd(feat, rcm, pxpy)
    fm = hip.denselize_fwd(torch.from_numpy(feat).cuda(), cm, tp).cpu().numpy()
    assert np.abs(fm - rfm).max() <= 1e-5 * np.abs(rfm).max()
    rgb = ref_rangelib.denselize_bwd(g, rcm, pxpy)
    gb = hip.denselize_bwd(torch.from_numpy(g).cuda(), cm, tp).cpu().numpy()
    assert np.abs(gb - rgb).max() <= 1e-6 * np.abs(rgb).max()


@pytest.mark.gpu
def test_hip_backend_shim_matches_reference_golden(hip, golden):
    """The `torchsparse.backend.*_cuda` surface (B-B) reproduces the reference's conv goldens, incl. the
    transposed call that needs a per-offset re-sort of the caller's map."""
    from openpcseg_amd import backend_shim as tb
    dev = "cuda"
    for tag, name, transposed in [("conv_k3s1_N", "k3s1", False), ("conv_k2s2_N", "k2s2", False), ("conv_k2s2_T", "k2s2", True)]:
        nbmaps = torch.from_numpy(golden["kmap_%s_nbmaps" % name]).int().to(dev)
        nbsizes = torch.from_numpy(golden["kmap_%s_nbsizes" % name]).int()  # on the CPU, like the reference passes it
        x, w, gy = (torch.from_numpy(golden[tag + s]).to(dev) for s in ("_x", "_w", "_gy"))
        y = torch.zeros(golden[tag + "_y"].shape, device=dev)
        tb.convolution_forward_cuda(x, y, w, nbmaps, nbsizes, transposed)
        assert np.abs(y.cpu().numpy() - golden[tag + "_y"]).max() <= 2e-5 * np.abs(golden[tag + "_y"]).max()
        gx, gw = torch.zeros_like(x), torch.zeros_like(w)
        tb.convolution_backward_cuda(x, gx, gy, w, gw, nbmaps, nbsizes, transposed)
        assert np.abs(gx.cpu().numpy() - golden[tag + "_gx"]).max() <= 2e-5 * np.abs(golden[tag + "_gx"]).max()
        assert np.abs(gw.cpu().numpy() - golden[tag + "_gw"]).max() <= 2e-5 * np.abs(golden[tag + "_gw"]).max()
    h = tb.hash_cuda(torch.from_numpy(golden["scene_coords"]).to(dev))
    q = torch.from_numpy(golden["query_q"]).to(dev)
    r = tb.hash_query_cuda(q, h, torch.arange(h.numel(), device=dev)) - 1
    assert np.array_equal(r.cpu().numpy(), golden["query_out"])


def test_scatter_max_result_is_tuple_like(oracle_backend):
    """torch_scatter.scatter_max returns (out, argmax): the wrapper's result unpacks, indexes and has length 2 like a
    tuple, and hands out an int64 argmax (materialised from the kernel's int32 one only when read)."""
    from openpcseg_amd.scatter import ScatterMaxResult, scatter_max
    src = torch.tensor([[1., 5.], [3., 2.], [0., 9.], [7., 7.]])
    idx = torch.tensor([1, 1, 3, 1])
    res = scatter_max(src, idx, dim=0)
    assert isinstance(res, ScatterMaxResult) and len(res) == 2
    out, arg = res
    assert out.tolist() == [[0., 0.], [7., 7.], [0., 0.], [0., 9.]]
    assert arg.dtype == torch.int64 and arg.tolist() == [[-1, -1], [3, 3], [-1, -1], [2, 2]]
    assert res[0] is out and res[1] is arg and res[-1] is arg and res[-2] is out
    assert scatter_max(src, idx, dim=0, dim_size=6)[0].shape == (6, 2)
    with pytest.raises(AssertionError):
        scatter_max(src, idx, dim=1)
    # half inputs (the cylinder front-end under autocast): same dtype out, exact, gradient in the input's dtype
    hsrc = src.to(torch.bfloat16).requires_grad_(True)
    hout, harg = scatter_max(hsrc, idx, dim=0)
    assert hout.dtype == torch.bfloat16 and hout.float().tolist() == out.tolist() and harg.tolist() == arg.tolist()
    hout.sum().backward()
    assert hsrc.grad.dtype == torch.bfloat16 and hsrc.grad.float().tolist() == [[0., 0.], [0., 0.], [1., 1.], [1., 1.]]


def test_kmap_entry_reads_like_the_reference_list(golden, oracle_backend):
    """kmaps[key] = [nbmaps, nbsizes, (n_in, n_out)] (TS:torchsparse/nn/functional/conv.py:174-176): index, unpack, len."""
    from openpcseg_amd import functional as F
    c = torch.from_numpy(golden["scene_coords"])
    entry = F.build_kernel_map(c, c, (3, 3, 3), (1, 1, 1), (1, 1, 1))
    assert len(entry) == 3
    nbmaps, nbsizes, sizes = entry
    assert np.array_equal(nbmaps.numpy(), golden["kmap_k3s1_nbmaps"]) and np.array_equal(nbsizes.numpy(), golden["kmap_k3s1_nbsizes"])
    assert sizes == (c.shape[0], c.shape[0]) and entry[0] is nbmaps and entry[-1] == sizes and entry[1] is nbsizes


def test_point_to_range_matches_the_reference_function(oracle_backend, monkeypatch):
    """rangelib.point_to_range (no host-to-device copy per call) against the reference's own `point_to_range`
    (R:pcseg/model/segmentor/fusion/rpvnet/rpvnet.py:73-91) run on CPU over the same backend: identical integer pixels -> identical maps."""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tests", "golden"))
    from stage_reference import reference_root
    if reference_root() is None:
        pytest.skip("reference sources not present")
    import make_golden as mg
    import openpcseg_amd
    from openpcseg_amd import rangelib
    openpcseg_amd.install_reference_aliases()
    mod = mg.import_reference_model("pcseg.model.segmentor.fusion.rpvnet.rpvnet")
    mod.rnf = sys.modules["range_utils.nn.functional"]
    monkeypatch.setattr(torch.Tensor, "cuda", lambda self, *a, **k: self)
    rng = np.random.default_rng(5)
    n, b, c = 4000, 2, 8
    pxpy = torch.from_numpy(np.concatenate([np.sort(rng.integers(0, b, n))[:, None].astype(np.float32),
                                            rng.uniform(-1, 1, size=(n, 2)).astype(np.float32)], 1))
    pxpy[:6, 1:] = torch.tensor([[-1.0, -1.0], [1.0, 1.0], [0.0, 0.0], [1.0, -1.0], [0.999999, 0.5], [-0.999999, -0.5]])
    pf = torch.from_numpy(rng.normal(size=(n, c)).astype(np.float32))
    for h, w in ((64, 2048), (16, 512), (4, 128)):
        ref = mod.point_to_range(pf, pxpy, b, h, w)
        got = rangelib.point_to_range(pf, pxpy, b, h, w)
        assert got.shape == (b, c, h, w) and torch.equal(got, ref), (h, w)
