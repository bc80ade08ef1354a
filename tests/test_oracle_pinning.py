"""Pin the oracle (oracle/pcs_oracle.c + oracle/oracle.py) to the REFERENCE.

Two anchors:
  1. tests/golden/ops_golden.npz -- outputs of the reference's own Python + compiled CPU backend
     (generated in-container by tests/golden/make_golden.py; travels to the GPU box);
  2. the reference's compiled CPU backend itself (oracle/_ref), called live where it exists.
Integer ops: bit-exact. Floating point: 1e-5 relative (fp32 summation order differs between
a BLAS GEMM and the scalar restatement).
"""
import os
import numpy as np
import pytest
import torch

from oracle import oracle as orc

RTOL = 1e-5


def close(a, b, rtol=RTOL):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    scale = max(np.abs(b).max(), 1e-6)
    assert a.shape == b.shape
    assert np.abs(a - b).max() <= rtol * scale + 1e-7, np.abs(a - b).max() / scale


def test_hash_known_answers(golden):
    # SURVEY.md section 2.2 K1 golden values (measured on the reference)
    kat = np.array([[0, 0, 0, 0], [1, 2, 3, 0], [-1, 5, 7, 1], [100, 200, 30, 3]], dtype=np.int32)
    expect = np.array([947293587111810033, 1043245732202901914, 348679674271016180, 909960678293697641])
    assert (orc.sphash(kat) == expect).all()
    assert (golden["kat_hash"] == expect).all()


def test_hash_and_kernel_hash(golden):
    assert (orc.sphash(golden["hash_coords"]) == golden["hash_out"]).all()
    assert (orc.sphash(golden["hash_coords"], golden["khash_offsets"]) == golden["khash_out"]).all()
    assert orc.sphash(golden["hash_coords"]).max() < 2 ** 60


def test_hash_query(golden):
    ref_h = orc.sphash(golden["scene_coords"])
    assert (orc.sphashquery(golden["query_q"], ref_h) == golden["query_out"]).all()
    assert (orc.sphashquery(np.zeros((0,), np.int64), ref_h).shape == (0,))
    # duplicates: first reference wins (query_cpu.cpp:22-26)
    assert orc.sphashquery(np.array([5, 7]), np.array([7, 5, 7, 5])).tolist() == [1, 0]


@pytest.mark.parametrize("name,args", [("k2s2", (2, 2, 1)), ("k3s2", (2, 3, 1)), ("k3s221", ((2, 2, 1), 3, 1))])
def test_downsample(golden, name, args):
    out = orc.spdownsample(golden["scene_coords"], *args)
    assert out.dtype == np.int32 and (out == golden["ds_" + name]).all()


def test_downsample_level2(golden):
    assert (orc.spdownsample(golden["ds_k2s2"], 2, 2, 2) == golden["ds_k2s2_l2"]).all()


@pytest.mark.parametrize("name,ks,in_stride", [("k3s1", 3, 1), ("k2s2", 2, 1), ("k133", (1, 3, 3), 1),
                                                ("k313", (3, 1, 3), 1), ("k3s2", 3, 1), ("k3s221", 3, 1)])
def test_kmap_order(golden, name, ks, in_stride):
    inc = golden["scene_coords"]
    outc = golden["ds_" + name] if ("ds_" + name) in golden.files else inc
    nbmaps, nbsizes = orc.build_kmap(inc, outc, ks, in_stride)
    assert (nbsizes == golden["kmap_%s_nbsizes" % name]).all()
    assert (nbmaps == golden["kmap_%s_nbmaps" % name]).all()


def test_kmap_level2(golden):
    c2 = golden["ds_k2s2"]
    nbmaps, nbsizes = orc.build_kmap(c2, c2, 3, 2)
    assert (nbmaps == golden["kmap_k3s1_l2_nbmaps"]).all() and (nbsizes == golden["kmap_k3s1_l2_nbsizes"]).all()


@pytest.mark.parametrize("tag,name,transposed", [("conv_k3s1_N", "k3s1", False), ("conv_k2s2_N", "k2s2", False),
                                                 ("conv_k2s2_T", "k2s2", True)])
def test_conv_fwd_bwd(golden, tag, name, transposed):
    nbmaps, nbsizes = golden["kmap_%s_nbmaps" % name], golden["kmap_%s_nbsizes" % name]
    n_in = golden["scene_coords"].shape[0]
    n_out = golden["ds_" + name].shape[0] if ("ds_" + name) in golden.files else n_in
    x, w, gy = golden[tag + "_x"], golden[tag + "_w"], golden[tag + "_gy"]
    y = orc.conv_fwd(x, w, nbmaps, nbsizes, (n_in, n_out), transposed)
    close(y, golden[tag + "_y"])
    gx, gw = orc.conv_bwd(x, gy, w, nbmaps, nbsizes, transposed)
    close(gx, golden[tag + "_gx"])
    close(gw, golden[tag + "_gw"])


def test_voxelize_devoxelize_tiweights(golden):
    out = orc.voxelize_fwd(golden["vox_feats"], golden["vox_idx"], golden["vox_counts"])
    close(out, golden["vox_out"], 1e-6)
    close(orc.voxelize_bwd(golden["vox_out"], golden["vox_idx"], golden["vox_counts"], 800), golden["vox_bwd"], 1e-6)
    assert (orc.spcount(golden["vox_idx"], 300) == golden["vox_counts"]).all()
    for s in (1, 2, 4):
        close(orc.calc_ti_weights(golden["tiw_coords"], golden["tiw_idxq"], s), golden["tiw_s%d" % s], 2e-6)
    w8 = np.ascontiguousarray(golden["tiw_s2"].T)
    idx8 = np.ascontiguousarray(golden["tiw_idxq"].T).astype(np.int32)
    close(orc.devoxelize_fwd(golden["devox_feat"], idx8, w8), golden["devox_out"], 1e-6)


def test_devoxelize_bwd_is_adjoint_of_fwd(golden):
    """devoxelize_cuda.cu:37-57: bwd is the transpose of the fwd gather (<fwd(f), g> == <f, bwd(g)>)."""
    rng = np.random.default_rng(0)
    w8 = np.ascontiguousarray(golden["tiw_s2"].T)
    idx8 = np.ascontiguousarray(golden["tiw_idxq"].T).astype(np.int32)
    f = rng.normal(size=(300, 5)).astype(np.float32)
    g = rng.normal(size=(800, 5)).astype(np.float32)
    lhs = (orc.devoxelize_fwd(f, idx8, w8).astype(np.float64) * g).sum()
    rhs = (f.astype(np.float64) * orc.devoxelize_bwd(g, idx8, w8, 300)).sum()
    assert abs(lhs - rhs) <= 1e-4 * max(abs(lhs), 1.0)


# ---- live against the reference's compiled backend (in-container only) -----------------------------
def test_live_reference_backend(ref_backend):
    rng = np.random.default_rng(11)
    coords = np.concatenate([rng.integers(-50, 50, size=(3000, 3)), np.zeros((3000, 1))], axis=1).astype(np.int32)
    coords = np.unique(coords, axis=0)
    tc = torch.from_numpy(coords)
    assert (ref_backend.hash(tc).numpy() == orc.sphash(coords)).all()
    off = orc.get_kernel_offsets(3)
    # single batch -> the CPU twin's kernel_hash is sound
    assert (ref_backend.ref.kernel_hash_cpu(tc, torch.from_numpy(off)).numpy() == orc.sphash(coords, off)).all()
    nbmaps, nbsizes = orc.build_kmap(coords, coords, 3)
    n = coords.shape[0]
    x = rng.normal(size=(n, 12)).astype(np.float32)
    w = (rng.normal(size=(27, 12, 20)) * 0.1).astype(np.float32)
    gy = rng.normal(size=(n, 20)).astype(np.float32)
    out = torch.zeros(n, 20)
    ref_backend.ref.convolution_forward_cpu(torch.from_numpy(x), out, torch.from_numpy(w),
                                            torch.from_numpy(nbmaps.astype(np.int32)),
                                            torch.from_numpy(nbsizes.astype(np.int32)), False)
    close(orc.conv_fwd(x, w, nbmaps, nbsizes, (n, n)), out.numpy())
    gin, gw = torch.zeros(n, 12), torch.zeros(27, 12, 20)
    ref_backend.ref.convolution_backward_cpu(torch.from_numpy(x), gin, torch.from_numpy(gy), torch.from_numpy(w), gw,
                                             torch.from_numpy(nbmaps.astype(np.int32)),
                                             torch.from_numpy(nbsizes.astype(np.int32)), False)
    ogx, ogw = orc.conv_bwd(x, gy, w, nbmaps, nbsizes)
    close(ogx, gin.numpy())
    close(ogw, gw.numpy())


@pytest.mark.parametrize("case", ["scan", "aniso", "ints"])
def test_sparse_quantize_restatement_matches_reference(case):
    """oracle.sparse_quantize (no np.unique; float64 floor-divide) == the reference's sparse_quantize with
    return_index / return_inverse on float points, anisotropic voxels and negative integer coordinates."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "quantize_golden.npz"))
    vox, idx, inv = orc.sparse_quantize(g[case + "_in"], g[case + "_vs"] if g[case + "_vs"].ndim else (float(g[case + "_vs"]),) * 3)
    assert np.array_equal(vox, g[case + "_vox"]) and np.array_equal(idx, g[case + "_idx"]) and np.array_equal(inv, g[case + "_inv"])


def _lovasz_golden():
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    from make_golden import lovasz_cases
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lovasz_golden.npz"))
    return [(name, p, lab, ign, float(g[name + "_loss"]), g[name + "_grad"]) for name, p, lab, ign in lovasz_cases()]


def test_lovasz_restatement_matches_reference():
    """oracle.lovasz_softmax (float64, stable sort) against the reference's lovasz_softmax run on CPU
    (tests/golden/lovasz_golden.npz, `make_golden.py lovasz`): ignore inside / outside the class range / none, absent
    classes, saturated probabilities; and the workload's two torch forms against the same vectors."""
    from openpcseg_amd.workloads.losses import lovasz_softmax, lovasz_softmax_per_class
    from seeded import lovasz_grad_mismatch
    for name, p, lab, ign, loss, grad in _lovasz_golden():
        lo, go = orc.lovasz_softmax(p, lab, ign)
        assert abs(lo - loss) <= 2e-6, (name, lo, loss)
        assert lovasz_grad_mismatch(p, lab, ign, go, grad) <= 2e-7, name   # (CPU torch.sort orders ties its own way)
        for fn in (lovasz_softmax, lovasz_softmax_per_class):
            tp = torch.from_numpy(p).requires_grad_(True)
            lt = fn(tp, torch.from_numpy(lab), ignore=ign)
            gt, = torch.autograd.grad(lt, tp)
            assert abs(float(lt.detach()) - loss) <= 2e-6 and lovasz_grad_mismatch(p, lab, ign, gt.numpy(), grad) <= 2e-7, (name, fn.__name__)
