"""BASELINE config 1: the pure-PyTorch CPU path (openpcseg_amd/cpu_fallback.py) against the reference's goldens
(tests/golden/ops_golden.npz: outputs of the reference's Python + compiled CPU backend) and against the oracle where
the goldens hold no vector. CPU only; the path is explicit opt-in and refuses device tensors."""
import numpy as np
import pytest
import torch

from openpcseg_amd import cpu_fallback, native
from openpcseg_amd import functional as F
from openpcseg_amd.sparse import SparseTensor


@pytest.fixture()
def be():
    with cpu_fallback.enabled() as b:
        yield b


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def test_opt_in_only():
    """Without install() the process-wide backend is the HIP one (and refuses CPU tensors); enabled() restores it."""
    before = native._BACKEND
    with cpu_fallback.enabled() as b:
        assert native.backend() is b and b.name == "torch-cpu"
    assert native._BACKEND is before
    assert "oracle" not in open(cpu_fallback.__file__).read().split('"""', 2)[2]  # the module body never touches oracle/


def test_hash_known_answers_and_goldens(golden, be):
    assert torch.equal(be.hash(t(golden["kat_coords"])), t(golden["kat_hash"]))
    assert torch.equal(be.hash(t(golden["hash_coords"])), t(golden["hash_out"]))
    assert torch.equal(be.kernel_hash(t(golden["hash_coords"]), t(golden["khash_offsets"])), t(golden["khash_out"]))
    neg = torch.tensor([[-1, -2, -3, 0], [2 ** 31 - 1, -(2 ** 31), 5, 7]], dtype=torch.int32)
    from oracle import oracle as orc
    assert (be.hash(neg).numpy() == orc.sphash(neg.numpy())).all()


def test_query_count_and_rulebooks(golden, be):
    coords = t(golden["scene_coords"])
    h = be.hash(coords)
    assert torch.equal(be.hash_query(t(golden["query_q"]), h), t(golden["query_out"]))
    dup = torch.tensor([5, 7, 5, 9], dtype=torch.int64)
    assert be.hash_query(torch.tensor([5, 9, 4]), dup).tolist() == [0, 3, -1]   # the first of equal references wins
    assert be.count(torch.tensor([0, 2, 2, -1, 1], dtype=torch.int32), 4).tolist() == [1, 1, 2, 0]
    for name, ks, st in [("k3s1", 3, 1), ("k2s2", 2, 2), ("k133", (1, 3, 3), 1), ("k313", (3, 1, 3), 1), ("k3s2", 3, 2),
                         ("k3s221", 3, (2, 2, 1))]:
        inp = SparseTensor(torch.zeros(coords.shape[0], 4), coords, 1)
        ks3 = (ks,) * 3 if isinstance(ks, int) else ks
        st3 = (st,) * 3 if isinstance(st, int) else st
        out = F.conv3d(inp, torch.zeros(int(np.prod(ks3)), 4, 4), ks3, stride=st3)
        entry = inp.kmaps[((1, 1, 1), ks3, st3, (1, 1, 1))]
        assert (entry[0].long().numpy() == golden["kmap_%s_nbmaps" % name]).all(), name
        assert (entry[1].numpy() == golden["kmap_%s_nbsizes" % name]).all(), name
        if "ds_" + name in golden.files:
            assert (out.C.numpy() == golden["ds_" + name]).all(), name


@pytest.mark.parametrize("tag,ks,stride,transposed", [("conv_k3s1_N", 3, 1, False), ("conv_k2s2_N", 2, 2, False),
                                                      ("conv_k2s2_T", 2, 2, True)])
def test_conv_forward_backward_vs_reference(golden, be, tag, ks, stride, transposed):
    coords = t(golden["scene_coords"])
    x = t(golden[tag + "_x"]).requires_grad_(True)
    w = t(golden[tag + "_w"]).requires_grad_(True)
    if not transposed:
        out = F.conv3d(SparseTensor(x, coords, 1), w, ks, stride=stride)
    else:
        fine = SparseTensor(torch.zeros(coords.shape[0], 8), coords, 1)
        fine.cmaps[(1, 1, 1)] = coords
        down = F.conv3d(fine, torch.zeros(8, 8, 12), ks, stride=stride)
        inp = SparseTensor(x, down.C, down.s)
        inp.cmaps, inp.kmaps = down.cmaps, down.kmaps
        out = F.conv3d(inp, w, ks, stride=stride, transposed=True)
    assert np.allclose(out.F.detach().numpy(), golden[tag + "_y"], rtol=1e-4, atol=1e-5)
    out.F.backward(t(golden[tag + "_gy"]))
    assert np.allclose(x.grad.numpy(), golden[tag + "_gx"], rtol=1e-4, atol=1e-5)
    assert np.allclose(w.grad.numpy(), golden[tag + "_gw"], rtol=1e-4, atol=1e-4)


def test_point_voxel_ops_vs_reference(golden, be):
    f = t(golden["vox_feats"]).requires_grad_(True)
    out = F.spvoxelize(f, t(golden["vox_idx"]), t(golden["vox_counts"]))
    assert np.allclose(out.detach().numpy(), golden["vox_out"], rtol=1e-5, atol=1e-6)
    out.backward(t(golden["vox_out"]))   # the fixture's backward ran on the forward output
    assert np.allclose(f.grad.numpy(), golden["vox_bwd"], rtol=1e-5, atol=1e-6)
    for s in (1, 2, 4):
        w = F.calc_ti_weights(t(golden["tiw_coords"]), t(golden["tiw_idxq"]), scale=s)
        assert np.allclose(w.numpy(), golden["tiw_s%d" % s], rtol=1e-5, atol=1e-6), s
    w1 = t(golden["tiw_s2"]).t().contiguous()   # the fixture devoxelised with the scale-2 weights
    idx = t(golden["tiw_idxq"]).t().contiguous()
    feat = t(golden["devox_feat"]).requires_grad_(True)
    out = F.spdevoxelize(feat, idx, w1)
    assert np.allclose(out.detach().numpy(), golden["devox_out"], rtol=1e-5, atol=1e-6)
    from oracle import oracle as orc
    gy = torch.randn(out.shape, generator=torch.Generator().manual_seed(3))
    out.backward(gy)
    assert np.allclose(feat.grad.numpy(), orc.devoxelize_bwd(gy.numpy(), idx.int().numpy(), w1.numpy(), feat.shape[0]),
                       rtol=1e-5, atol=1e-6)


def test_quantize_scatter_and_range_ops_vs_oracle(golden, be):
    from oracle import oracle as orc
    for b in range(2):
        vox, idx, inv = be.quantize(t(golden["quant_in_%d" % b]), (1, 1, 1), True, True)
        assert (vox.numpy() == golden["quant_out_%d" % b]).all() and (idx.numpy() == golden["quant_idx_%d" % b]).all()
        assert (vox[inv].numpy() == golden["quant_in_%d" % b]).all()
    rng = np.random.default_rng(5)
    src = rng.normal(size=(400, 6)).astype(np.float32)
    index = rng.integers(0, 50, size=400).astype(np.int64)
    index[index == 7] = 8   # an empty row
    out, arg = be.scatter_max_fwd(t(src), t(index), 50)
    o_out, o_arg = orc.scatter_max(src, index, 50)
    assert np.array_equal(out.numpy(), o_out) and np.array_equal(arg.numpy(), o_arg)
    g = rng.normal(size=(50, 6)).astype(np.float32)
    assert np.allclose(be.scatter_max_bwd(t(g), arg, 400).numpy(), orc.scatter_max_bwd(g, o_arg, 400))
    pxpy = np.stack([rng.integers(0, 2, 300), rng.integers(-1, 17, 300), rng.integers(-1, 9, 300)], 1).astype(np.int32)
    cm = be.map_count(t(pxpy), 2, 8, 16)
    assert np.array_equal(cm.numpy(), orc.map_count(pxpy, 2, 8, 16))
    feat = rng.normal(size=(300, 4)).astype(np.float32)
    dense = be.denselize_fwd(t(feat), cm, t(pxpy))
    assert np.allclose(dense.numpy(), orc.denselize_fwd(feat, cm.numpy(), pxpy), rtol=1e-5, atol=1e-6)
    gd = rng.normal(size=dense.shape).astype(np.float32)
    assert np.allclose(be.denselize_bwd(t(gd), cm, t(pxpy)).numpy(), orc.denselize_bwd(gd, cm.numpy(), pxpy), rtol=1e-5, atol=1e-6)


def test_refuses_device_tensors(be):
    class FakeCuda(torch.Tensor):
        is_cuda = True
    x = torch.zeros(2, 4, dtype=torch.int32).as_subclass(FakeCuda)
    with pytest.raises(RuntimeError, match="CPU tensor"):
        be.hash(x)


def test_prebuild_coords_matches_the_lazy_levels(golden, be):
    """functional.prebuild_coords (host logic, device-independent): the levels it leaves in cmaps are the ones the strided
    convolutions compute themselves -- both spdownsample branches, anisotropic strides, levels already present are kept."""
    coords = t(golden["scene_coords"])
    steps = [(2, 2), ((2, 2, 1), 3), (2, 2)]
    ws = [torch.zeros(8, 4, 4), torch.zeros(27, 4, 4), torch.zeros(8, 4, 4)]

    def run(prebuild):
        x = SparseTensor(torch.zeros(coords.shape[0], 4), coords, 1)
        if prebuild:
            F.prebuild_coords(x, steps)
            assert sorted(x.cmaps) == [(2, 2, 2), (4, 4, 2), (8, 8, 4)]
            marker = x.cmaps[(2, 2, 2)]
            F.prebuild_coords(x, steps[:1])
            assert x.cmaps[(2, 2, 2)] is marker
        for w, (s, k) in zip(ws, steps):
            x = F.conv3d(x, w, k, stride=s)
        return x

    a, b = run(False), run(True)
    assert sorted(a.cmaps) == sorted(b.cmaps) and all(torch.equal(a.cmaps[k], b.cmaps[k]) for k in a.cmaps)
    assert (a.cmaps[(2, 2, 2)].numpy() == golden["ds_k2s2"]).all()


def test_prebuild_encoder_reads_the_models_own_stage_convolutions(golden, be):
    """block_fusion._prebuild_encoder (host logic): the plan comes from the FIRST Conv3d of stage1..stage4 -- its own stride and
    kernel size --, any other layout (a stage that does not open with a strided, non-transposed convolution, a missing stage)
    prebuilds nothing, and PCS_PREBUILD_LEVELS=0 switches it off."""
    import os
    from torch import nn
    from openpcseg_amd import block_fusion as bf
    from openpcseg_amd.modules import Conv3d

    def model(specs):
        m = nn.Module()
        for i, (k, s, tr) in enumerate(specs):
            setattr(m, "stage%d" % (i + 1), nn.Sequential(nn.Sequential(Conv3d(4, 4, kernel_size=k, stride=s, transposed=tr)), nn.Identity()))
        return m

    coords = t(golden["scene_coords"])
    x = SparseTensor(torch.zeros(coords.shape[0], 4), coords, 1)
    good = model([(2, 2, False), (3, 2, False), (2, 2, False), (2, (2, 2, 1), False)])
    bf._prebuild_encoder(good, x)
    assert good.__dict__["_pcs_enc_steps"] == [((2, 2, 2), (2, 2, 2)), ((2, 2, 2), (3, 3, 3)), ((2, 2, 2), (2, 2, 2)), ((2, 2, 1), (2, 2, 2))]
    assert sorted(x.cmaps) == [(2, 2, 2), (4, 4, 4), (8, 8, 8), (16, 16, 8)]
    assert (x.cmaps[(2, 2, 2)].numpy() == golden["ds_k2s2"]).all()
    for bad in (model([(2, 2, False), (3, 1, False), (2, 2, False), (2, 2, False)]), model([(2, 2, False), (2, 2, True), (2, 2, False), (2, 2, False)]),
                model([(2, 2, False)] * 3)):
        y = SparseTensor(torch.zeros(coords.shape[0], 4), coords, 1)
        bf._prebuild_encoder(bad, y)
        assert bad.__dict__["_pcs_enc_steps"] == [] and not y.cmaps
    os.environ["PCS_PREBUILD_LEVELS"] = "0"
    try:
        z = SparseTensor(torch.zeros(coords.shape[0], 4), coords, 1)
        bf._prebuild_encoder(good, z)
        assert not z.cmaps
    finally:
        del os.environ["PCS_PREBUILD_LEVELS"]
