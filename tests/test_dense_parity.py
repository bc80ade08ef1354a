"""Parity of the hot kernels at BASELINE density (-m gpu): the resolution levels of FULL 120k-point scans.

The op-level cases of test_hip_parity.py run on a 3k-voxel scene with 1.5 pairs per output row, where an offset rarely
has two 16-row blocks in a tile. Here the maps are the ones bench.py times: two synthetic frames (seeds 0, 1) at
0.05 m, levels at tensor strides 1..8 with 4..8.8 pairs per row, so the row-block groups (R = 2), the multi-wave
ticket queue, the 8-wave / 384-row configuration and the 64-column tiles of conv_os5_kernel and the split-reduction
wgrad2_kernel all carry real work -- and every result is compared with the scalar oracle
(oracle/pcs_oracle.c::orc_conv_fwd / orc_conv_bwd <- TS:torchsparse/backend/convolution/convolution_cpu.cpp:38-183).
Rulebooks are compared bit-exactly, order included, on the full frame and on the 12-frame batch of the bench
(oracle.build_kmap <- TS:torchsparse/nn/functional/conv.py:156-176).
Tolerance for fp32: 2e-5 of the tensor maximum (MFMA vs scalar summation order), as in test_hip_parity.py."""
import os
import numpy as np
import pytest
import torch

from oracle import oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda"


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def close(a, b, rtol):
    a = a.detach().cpu().numpy().astype(np.float64) if isinstance(a, torch.Tensor) else np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    err = np.abs(a - b).max() / max(np.abs(b).max(), 1e-6)
    assert err <= rtol, err


def scan_levels(seeds):
    """{tensor stride: (N,4) int32 coords} of a batch of full scans, built by the ORACLE: stride 1 in ascending-hash
    order (what initial_voxelize hands to the first conv, R:.../minkunet/utils.py:11-36), coarser levels in the
    lexicographic order spdownsample produces (TS:.../functional/downsample.py:47-51)."""
    from openpcseg_amd.workloads.synthetic import make_batch
    c = make_batch(list(seeds))["lidar"].C.numpy()
    h = orc.sphash(c)
    lv = {1: np.ascontiguousarray(c[np.argsort(h, kind="stable")])}
    for s in (1, 2, 4):
        lv[2 * s] = orc.spdownsample(lv[s], 2, 2, s)
    return lv


@pytest.fixture(scope="module")
def levels():
    return scan_levels([0, 1])


_MAPS = {}


def level_map(levels, stride):
    """(HIP KmapEntry, oracle nbmaps, nbsizes, n) of the k = 3 submanifold map of one level; the HIP rulebook is
    asserted bit-equal to the oracle's, order included, before it is used."""
    if stride not in _MAPS:
        from openpcseg_amd import functional as F
        c = levels[stride]
        dc = t(c)
        entry = F.build_kernel_map(dc, dc, (3, 3, 3), (stride,) * 3, (1, 1, 1))
        nbmaps, nbsizes = orc.build_kmap(c, c, 3, stride)
        assert np.array_equal(entry[1].cpu().numpy(), nbsizes)
        assert np.array_equal(entry[0].cpu().numpy().astype(np.int64), nbmaps)
        _MAPS[stride] = (entry, nbmaps, nbsizes, c.shape[0])
    return _MAPS[stride]


def test_levels_have_baseline_density(levels):
    """P/N = 3.95 / 5.72 / 8.55 / 8.83 at strides 1 / 2 / 4 / 8 (one frame; two here)."""
    n = {s: levels[s].shape[0] for s in levels}
    assert n[1] > 180000 and n[4] > 50000 and n[8] > 17000
    for s, lo in ((1, 3.5), (2, 5.0), (4, 7.5), (8, 7.5)):
        nbsizes = orc.build_kmap(levels[s], levels[s], 3, s)[1]
        assert nbsizes.sum() / n[s] > lo, (s, nbsizes.sum() / n[s])


# stride, cin, cout, forced tile height (None = the per-layer pick of pcs_conv_pick_tile_rows)
FWD_CASES = [
    (4, 128, 128, None), (4, 128, 128, 112), (4, 128, 128, 256), (4, 128, 128, 288),
    (4, 192, 128, None), (4, 192, 128, 128), (4, 256, 128, 224),
    (8, 256, 256, None), (8, 256, 256, 128), (8, 256, 256, 224), (8, 256, 256, 288),
    (8, 384, 256, None), (8, 384, 256, 256),
    (2, 64, 64, None), (2, 128, 64, 256), (2, 64, 64, 192), (4, 64, 32, 224), (4, 128, 128, 176), (8, 128, 256, 192),
    (1, 96, 96, 384), (1, 128, 96, None), (1, 96, 96, 128), (1, 96, 128, None),
]
# BASELINE config 5 widths (RPVNet mk34 cr 1.75: PLANES x 1.75 = 56/56/112/224/448/448/224/168/168, decoder concat
# inputs 672/336/224/224, R:pcseg/model/segmentor/fusion/rpvnet/rpvnet.py:211-213) and other cin % 32 != 0 shapes: the
# TAIL instances of conv_os5_kernel / conv_os5h_kernel (partial last contraction block, odd block counts, column
# tiles that pad 56 -> 64, 112 -> 128, 168 -> 192, 336 -> 384 columns)
TAIL_CASES = [(1, 56, 56, None), (2, 56, 112, None), (2, 112, 112, None), (1, 168, 168, None), (1, 224, 168, 256),
              (4, 336, 224, None), (8, 224, 448, None), (8, 672, 448, None), (8, 448, 448, 224),
              (2, 48, 48, None), (4, 100, 64, None), (4, 36, 24, 128), (8, 44, 200, None)]
FWD_CASES += TAIL_CASES


@pytest.mark.parametrize("stride,cin,cout,tile", FWD_CASES)
def test_conv_forward_dense_map(hip, levels, stride, cin, cout, tile):
    """conv_os5_kernel forward on a BASELINE-density map vs the oracle; the same launch twice is bit-identical."""
    entry, nbmaps, nbsizes, n = level_map(levels, stride)
    rng = np.random.default_rng(stride * 100000 + cin * 100 + cout)
    x = rng.normal(size=(n, cin)).astype(np.float32)
    w = (rng.normal(size=(27, cin, cout)) / np.sqrt(cin * 27)).astype(np.float32)
    dx, dw = t(x), t(w)
    y = hip.conv_gather_gemm(dx, dw, entry.fwd, tile_rows=tile)
    close(y, orc.conv_fwd(x, w, nbmaps, nbsizes, (n, n)), 2e-5)
    assert torch.equal(y, hip.conv_gather_gemm(dx, dw, entry.fwd, tile_rows=tile))


@pytest.mark.parametrize("stride,tile", [(1, 384), (2, 128), (4, 224), (8, 288), (8, 112)])
def test_tile_order_heaviest_first(hip, levels, stride, tile):
    """pcs_rulebook_tile_order: a permutation of the row tiles, work (16-row blocks over all offsets) non-increasing. With
    PCS_TILE_ORDER_XCD=1 (opt-in): the slots of XCD c (launch position % 8 == c) hold exactly the c-th contiguous eighth of the
    tiles (lengths differ by at most one), work non-increasing inside it."""
    import os
    entry = level_map(levels, stride)[0]
    km = entry.fwd
    seg = hip._segments(km, tile).view(27, -1).cpu().numpy().astype(np.int64)
    order = hip._tile_order(km, tile).cpu().numpy()
    ntiles = (km.n_dst + tile - 1) // tile
    assert order.shape == (ntiles,) and np.array_equal(np.sort(order), np.arange(ntiles))
    work = ((seg[:, 1:] - seg[:, :-1] + 15) // 16).sum(0)
    if os.environ.get("PCS_TILE_ORDER_XCD", "0") == "0":
        w = work[order]
        assert (w[:-1] >= w[1:]).all() and w[0] == work.max()
        return
    q, r = divmod(ntiles, 8)
    for c in range(8):
        lo, ln = c * q + min(c, r), q + (1 if c < r else 0)
        mine = order[c::8]
        assert np.array_equal(np.sort(mine), np.arange(lo, lo + ln))
        w = work[mine]
        assert (w[:-1] >= w[1:]).all() and (ln == 0 or w[0] == work[lo:lo + ln].max())


def test_tile_order_per_xcd_variant():
    """PCS_TILE_ORDER_XCD=1 (the opt-in order: heaviest first inside each XCD's contiguous eighth) is read once per process: the
    order property and the order-independence of the convolution results are checked in a subprocess started with it."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PCS_TILE_ORDER_XCD="1")
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_dense_parity.py"), "-q", "-x", "-m", "gpu",
                        "-k", "test_tile_order_heaviest_first or test_conv_is_independent_of_the_tile_order", "-p", "no:cacheprovider"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert p.returncode == 0 and " passed" in p.stdout, (p.stdout + p.stderr)[-1500:]


@pytest.mark.parametrize("stride,cin,cout,tile", [(4, 128, 128, None), (8, 256, 256, None), (1, 96, 96, 384), (2, 64, 64, 128)])
def test_conv_is_independent_of_the_tile_order(hip, levels, stride, cin, cout, tile):
    """The tile order only decides which workgroup slot runs which tile: outputs and the BatchNorm partial sums are
    bit-identical with the heaviest-first order and in row order, fp32 and half kernels."""
    entry = level_map(levels, stride)[0]
    n = entry.fwd.n_dst
    x = torch.randn(n, cin, device=DEV)
    w = torch.randn(27, cin, cout, device=DEV) * 0.05
    a, b = [], []
    ya = hip.conv_gather_gemm(x, w, entry.fwd, tile_rows=tile, bn_sums=a, ordered="force")
    yb = hip.conv_gather_gemm(x, w, entry.fwd, tile_rows=tile, bn_sums=b, ordered=False)
    assert torch.equal(ya, yb) and len(a) == len(b) and all(torch.equal(p, q) for p, q in zip(a, b))
    for dtype in (torch.bfloat16, torch.float16):
        xh, wp = x.to(dtype), hip.prepare_weights_h(w, dtype, transpose=False)
        assert torch.equal(hip.conv_gather_gemm_h(xh, wp, 27, cout, entry.fwd, tile_rows=tile, ordered="force"),
                           hip.conv_gather_gemm_h(xh, wp, 27, cout, entry.fwd, tile_rows=tile, ordered=False))


@pytest.mark.parametrize("stride,cin,cout", [(4, 128, 128), (4, 192, 128), (8, 256, 256), (8, 384, 256), (1, 128, 96),
                                             (2, 112, 56), (4, 336, 224), (1, 168, 168), (8, 672, 448), (4, 100, 36)])
def test_conv_backward_dense_map(hip, levels, stride, cin, cout):
    """dgrad (conv_os5_kernel on the input-sorted map, transposed weights) and wgrad (wgrad2_kernel + split reduction)
    vs orc_conv_bwd on a BASELINE-density map."""
    entry, nbmaps, nbsizes, n = level_map(levels, stride)
    rng = np.random.default_rng(stride * 100000 + cin * 100 + cout + 1)
    x = rng.normal(size=(n, cin)).astype(np.float32)
    gy = rng.normal(size=(n, cout)).astype(np.float32)
    w = (rng.normal(size=(27, cin, cout)) / np.sqrt(cin * 27)).astype(np.float32)
    ogx, ogw = orc.conv_bwd(x, gy, w, nbmaps, nbsizes)
    gx = hip.conv_gather_gemm(t(gy), hip.transpose_weights(t(w)), entry.rev)
    close(gx, ogx, 2e-5)
    gw = hip.conv_wgrad(t(x), t(gy), entry.fwd, 0)
    close(gw, ogw, 2e-5)
    assert torch.equal(gw, hip.conv_wgrad(t(x), t(gy), entry.fwd, 0))
    # the same gradient through the bf16 MFMAs (three bf16 planes per operand, six products): fp32-grade, same bound
    gws = hip.conv_wgrad(t(x), t(gy), entry.fwd, 0, split=True)
    close(gws, ogw, 2e-5)
    assert torch.equal(gws, hip.conv_wgrad(t(x), t(gy), entry.fwd, 0, split=True))
    # "fp32-grade": against the oracle's DOUBLE-accumulated gradient the split path errs at most twice what the fp32 MFMA does
    e32 = np.abs(gw.cpu().numpy().astype(np.float64) - ogw).max()
    e3 = np.abs(gws.cpu().numpy().astype(np.float64) - ogw).max()
    assert e3 <= 2.0 * e32 + 1e-7 * np.abs(ogw).max(), (e3, e32)


def test_strided_maps_full_frame_bit_exact(hip, levels):
    """k2 s2 down-conv maps (and the coordinates spdownsample emits) between the levels of the full scans, bit-exact
    incl. order; plus one strided + one transposed conv over them vs the oracle."""
    from openpcseg_amd import functional as F
    for s in (1, 2, 4):
        cin_c, cout_c = levels[s], levels[2 * s]
        out = hip.downsample(t(cin_c), [2 * s] * 3)
        assert np.array_equal(out.cpu().numpy(), cout_c)
        entry = F.build_kernel_map(t(cin_c), out, (2, 2, 2), (s,) * 3, (1, 1, 1))
        nbmaps, nbsizes = orc.build_kmap(cin_c, cout_c, 2, s)
        assert np.array_equal(entry[1].cpu().numpy(), nbsizes)
        assert np.array_equal(entry[0].cpu().numpy().astype(np.int64), nbmaps)
    # stride 4 -> 8, 128 -> 128 down-conv and its transposed up-conv 256 -> 128 (decoder shape)
    n_in, n_out = cin_c.shape[0], cout_c.shape[0]
    rng = np.random.default_rng(5)
    x = rng.normal(size=(n_in, 128)).astype(np.float32)
    w = (rng.normal(size=(8, 128, 128)) / np.sqrt(128 * 8)).astype(np.float32)
    close(hip.conv_gather_gemm(t(x), t(w), entry.fwd), orc.conv_fwd(x, w, nbmaps, nbsizes, (n_in, n_out)), 2e-5)
    xu = rng.normal(size=(n_out, 256)).astype(np.float32)
    wu = (rng.normal(size=(8, 256, 128)) / np.sqrt(256 * 8)).astype(np.float32)
    close(hip.conv_gather_gemm(t(xu), t(wu), entry.rev),
          orc.conv_fwd(xu, wu, nbmaps, nbsizes, (n_in, n_out), transposed=True), 2e-5)


def test_rulebook_bench_batch_bit_exact(hip):
    """The 12-frame batch bench.py times (about 1.16 M voxels, 4.6 M pairs): stride-1 k3 rulebook and the first
    down-conv map vs the oracle, bit-exact incl. order."""
    from openpcseg_amd import functional as F
    lv = scan_levels(range(12))
    c1 = lv[1]
    assert c1.shape[0] > 1_000_000 and int(c1[:, 3].max()) == 11
    d1 = t(c1)
    entry = F.build_kernel_map(d1, d1, (3, 3, 3), (1, 1, 1), (1, 1, 1))
    nbmaps, nbsizes = orc.build_kmap(c1, c1, 3, 1)
    assert np.array_equal(entry[1].cpu().numpy(), nbsizes)
    assert np.array_equal(entry[0].cpu().numpy().astype(np.int64), nbmaps)
    rev = entry.rev  # mirrored map == probing the negated offsets
    built = hip.build_kmap(d1, d1, -entry._ctx[2])
    assert torch.equal(rev.pairs, built.pairs) and rev.koff_host == built.koff_host
    out = hip.downsample(d1, [2, 2, 2])
    assert np.array_equal(out.cpu().numpy(), lv[2])
    e2 = F.build_kernel_map(d1, out, (2, 2, 2), (1, 1, 1), (1, 1, 1))
    nb2, ns2 = orc.build_kmap(c1, lv[2], 2, 1)
    assert np.array_equal(e2[1].cpu().numpy(), ns2) and np.array_equal(e2[0].cpu().numpy().astype(np.int64), nb2)


# ---- fp32 convolution on the bf16 MFMAs (three planes per operand, six products; opt-in) --------------------------------
X3_CASES = [(4, 128, 128, None), (4, 192, 128, 224), (8, 256, 256, None), (8, 384, 256, 256), (1, 96, 96, 384), (1, 96, 96, 192),
            (1, 128, 96, None), (2, 64, 64, None), (1, 32, 32, None), (2, 32, 64, 128), (4, 160, 96, 112), (8, 256, 32, None),
            (1, 56, 56, None), (2, 112, 112, None), (1, 168, 168, None), (4, 336, 224, None), (8, 672, 448, None), (2, 40, 56, None)]


@pytest.mark.parametrize("stride,cin,cout,tile", X3_CASES)
def test_conv_x3_forward_dense_map(hip, levels, stride, cin, cout, tile):
    """conv_os5x_kernel (fp32 in / out, operands as three bf16 planes, six plane products) vs the oracle at the fp32
    bound, bit-reproducible, with bias and BatchNorm partials; and fp32-GRADE: against a float64 evaluation its error is
    at most twice the fp32 MFMA kernel's (the same statement the three-plane wgrad is held to)."""
    entry, nbmaps, nbsizes, n = level_map(levels, stride)
    rng = np.random.default_rng(stride * 100000 + cin * 100 + cout + 3)
    x = rng.normal(size=(n, cin)).astype(np.float32)
    w = (rng.normal(size=(27, cin, cout)) / np.sqrt(cin * 27)).astype(np.float32)
    bias = rng.normal(size=cout).astype(np.float32)
    dx, dw = t(x), t(w)
    assert hip.conv_x3_applies(cin, cout, 27)
    wp = hip.prepare_weights_x3(dw, transpose=False)
    y = hip.conv_gather_gemm_x3(dx, wp, 27, cout, entry.fwd, tile_rows=tile)
    assert y.dtype == torch.float32 and y.shape == (n, cout)
    ref = orc.conv_fwd(x, w, nbmaps, nbsizes, (n, n))
    close(y, ref, 2e-5)
    assert torch.equal(y, hip.conv_gather_gemm_x3(dx, wp, 27, cout, entry.fwd, tile_rows=tile))
    got = []
    yb = hip.conv_gather_gemm_x3(dx, wp, 27, cout, entry.fwd, bias=t(bias), tile_rows=tile, bn_sums=got)
    close(yb, ref + bias[None, :], 2e-5)
    if got:
        yd = yb.double()
        assert torch.allclose(got[0][:cout], yd.sum(0), rtol=0, atol=1e-6 * float(yd.abs().sum(0).max()))
        assert torch.allclose(got[0][cout:2 * cout], (yd * yd).sum(0), rtol=1e-6)
    # fp32-grade: error against the double-precision result, on a row sample (torch fp64 index_add on the device)
    rows = torch.arange(0, n, 7, device=DEV)
    pairs = entry.fwd.pairs.long()
    koff = entry.fwd.koff_host
    y64 = torch.zeros(n, cout, dtype=torch.float64, device=DEV)
    x64, w64 = dx.double(), dw.double()
    for k in range(27):
        pk = pairs[koff[k]:koff[k + 1]]
        if pk.numel():
            y64.index_add_(0, pk[:, 1], x64[pk[:, 0]] @ w64[k])
    y32 = hip.conv_gather_gemm(dx, dw, entry.fwd, tile_rows=tile)
    e3 = float((y.double() - y64)[rows].abs().max())
    e32 = float((y32.double() - y64)[rows].abs().max())
    assert e3 <= 2.0 * e32 + 1e-7 * float(y64.abs().max()), (e3, e32)


@pytest.mark.parametrize("stride,cin,cout", [(4, 128, 128), (8, 384, 256), (1, 128, 96), (2, 112, 56), (4, 336, 224)])
def test_conv_x3_dgrad_dense_map(hip, levels, stride, cin, cout):
    """dgrad on the split kernel (weights prepared with transpose=True, input-sorted map) vs orc_conv_bwd."""
    entry, nbmaps, nbsizes, n = level_map(levels, stride)
    rng = np.random.default_rng(stride * 100000 + cin * 100 + cout + 5)
    x = rng.normal(size=(n, cin)).astype(np.float32)
    gy = rng.normal(size=(n, cout)).astype(np.float32)
    w = (rng.normal(size=(27, cin, cout)) / np.sqrt(cin * 27)).astype(np.float32)
    ogx, _ = orc.conv_bwd(x, gy, w, nbmaps, nbsizes)
    wpt = hip.prepare_weights_x3(t(w), transpose=True)
    gx = hip.conv_gather_gemm_x3(t(gy), wpt, 27, cin, entry.rev)
    close(gx, ogx, 2e-5)


def test_conv3d_policy_bf16x3(hip, levels):
    """functional.set_conv_policy('bf16x3'): conv3d forward and input gradient take the split kernel on the shapes it serves
    and agree with the fp32 MFMA path at the fp32 bound; the policy is off by default and restored here."""
    from openpcseg_amd import functional as F
    from openpcseg_amd.sparse import SparseTensor
    assert F.get_conv_policy() == "fp32"
    c = t(levels[4])
    g = torch.Generator(device=DEV).manual_seed(13)
    x = torch.randn(c.shape[0], 64, device=DEV, generator=g)
    w1 = (torch.randn(27, 64, 128, device=DEV, generator=g) * 0.03).requires_grad_(True)
    w2 = (torch.randn(27, 128, 16, device=DEV, generator=g) * 0.03).requires_grad_(True)   # 16 columns: stays on fp32 MFMA

    def run():
        for w in (w1, w2):
            w.grad = None
        xs = SparseTensor(x.clone().requires_grad_(True), c, 4)
        y = F.conv3d(F.conv3d(xs, w1, 3), w2, 3)
        y.F.square().sum().backward()
        return y.F.detach(), w1.grad.clone(), w2.grad.clone(), xs.F.grad.clone()
    base = run()
    F.set_conv_policy("bf16x3")
    try:
        split = run()
    finally:
        F.set_conv_policy("fp32")
    for a, b in zip(split, base):
        assert a.dtype == torch.float32 and (a - b).abs().max() <= 2e-5 * b.abs().max(), ((a - b).abs().max(), b.abs().max())
    assert not torch.equal(split[0], base[0])   # it really ran another kernel


# ---- half-precision path (bf16 / fp16 storage, 16-bit MFMA, fp32 accumulate) -------------------------------------
def _round_half(a, dtype):
    """fp32 array -> the nearest bf16 / fp16 values, as fp32 (what the kernels see as their operands)."""
    return torch.from_numpy(np.ascontiguousarray(a)).to(dtype).float().numpy()


# rounding of the stored output: half an ulp = 2^-9 (bf16, 8 significand bits) / 2^-12 (fp16, 11 bits) of the value, plus
# the fp32 accumulation-order term of the fp32 tests relative to the tensor maximum
_HALF_TOL = {torch.bfloat16: 2.0 ** -8, torch.float16: 2.0 ** -11}


def close_half(y, ref, dtype):
    y = y.detach().float().cpu().numpy().astype(np.float64)
    ref = np.asarray(ref, np.float64)
    assert y.shape == ref.shape
    bound = _HALF_TOL[dtype] * np.abs(ref) + 4e-5 * max(np.abs(ref).max(), 1e-6)
    bad = np.abs(y - ref) > bound
    assert not bad.any(), (int(bad.sum()), float(np.abs(y - ref).max()), float(np.abs(ref).max()))


HALF_CASES = [(4, 128, 128, None), (4, 192, 128, 128), (8, 256, 256, None), (8, 384, 256, 256), (1, 96, 96, 384),
              (1, 128, 96, None), (2, 64, 64, None), (4, 160, 96, 112), (8, 256, 20, None),
              # config 5 widths and other cin % 32 != 0 rows (cin % 8 == 0): TAIL instance, zero-padded weight fragments
              (1, 56, 56, None), (2, 112, 112, None), (1, 168, 168, None), (4, 336, 224, None), (8, 224, 448, None),
              (8, 672, 448, None), (4, 72, 24, 128), (2, 40, 56, None)]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("stride,cin,cout,tile", HALF_CASES)
def test_half_conv_forward_dense_map(hip, levels, dtype, stride, cin, cout, tile):
    """conv_os5h_kernel vs the oracle run on the SAME half-rounded operands in fp32 (the 16-bit MFMA multiplies
    exactly and accumulates in fp32, so only the output rounding separates the two); odd step counts (cin = 160, 96),
    partial column tiles (cout = 20, 96) and forced tile heights included; bit-reproducible."""
    entry, nbmaps, nbsizes, n = level_map(levels, stride)
    rng = np.random.default_rng(stride * 100000 + cin * 100 + cout + 7)
    x = _round_half(rng.normal(size=(n, cin)).astype(np.float32), dtype)
    w = _round_half((rng.normal(size=(27, cin, cout)) / np.sqrt(cin * 27)).astype(np.float32), dtype)
    bias = rng.normal(size=cout).astype(np.float32)
    wp = hip.prepare_weights_h(t(w), dtype, transpose=False)
    dx = t(x).to(dtype)
    y = hip.conv_gather_gemm_h(dx, wp, 27, cout, entry.fwd, tile_rows=tile)
    assert y.dtype == dtype and y.shape == (n, cout)
    ref = orc.conv_fwd(x, w, nbmaps, nbsizes, (n, n))
    close_half(y, ref, dtype)
    assert torch.equal(y, hip.conv_gather_gemm_h(dx, wp, 27, cout, entry.fwd, tile_rows=tile))
    yb = hip.conv_gather_gemm_h(dx, wp, 27, cout, entry.fwd, bias=t(bias), tile_rows=tile)
    close_half(yb, ref + bias[None, :], dtype)


@pytest.mark.parametrize("stride,cin,cout,tile", [(4, 128, 128, None), (1, 96, 96, None), (2, 32, 32, None), (8, 256, 256, None),
                                                  (4, 64, 48, 64), (2, 36, 44, None)])
def test_conv_write_back_addend(hip, levels, stride, cin, cout, tile):
    """pcs_conv_gather_gemm_f32_add / _h_add: dst = conv + bias + addend, the addend (a skip path's gradient in dgrad) added in the
    write-back. fp32: bit-identical to the plain launch plus one fp32 addition; halfs: added in fp32 before the ONE rounding."""
    entry, nbmaps, nbsizes, n = level_map(levels, stride)
    g = torch.Generator(device=DEV).manual_seed(stride * 1000 + cin + cout)
    x = torch.randn(n, cin, device=DEV, generator=g)
    w = torch.randn(27, cin, cout, device=DEV, generator=g) / np.sqrt(cin * 27)
    add = torch.randn(n, cout, device=DEV, generator=g)
    bias = torch.randn(cout, device=DEV, generator=g)
    assert hip.conv_supports_addend(cin, cout, 27, 0)
    y = hip.conv_gather_gemm(x, w, entry.fwd, tile_rows=tile)
    ya = hip.conv_gather_gemm(x, w, entry.fwd, tile_rows=tile, addend=add)
    assert torch.equal(ya, y + add)
    yb = hip.conv_gather_gemm(x, w, entry.fwd, tile_rows=tile, addend=add, bias=bias)
    assert torch.allclose(yb, y + bias + add, rtol=0, atol=2e-6 * float(y.abs().max()))
    if hip.conv_h_applies(cin, cout, 27):
        for dtype in (torch.bfloat16, torch.float16):
            assert hip.conv_supports_addend(cin, cout, 27, hip._HALF[dtype])
            wp = hip.prepare_weights_h(w, dtype, transpose=False)
            xh, ah = x.to(dtype), add.to(dtype)
            for mode in (0, 3):   # conv_os5h and the weight-stationary kernel's own epilogue
                try:
                    _ws_mode(hip, mode)
                    yh = hip.conv_gather_gemm_h(xh, wp, 27, cout, entry.fwd, tile_rows=tile, addend=ah)
                finally:
                    _ws_mode(hip, -1)
                ref = orc.conv_fwd(xh.float().cpu().numpy(), _round_half(w.cpu().numpy(), dtype), nbmaps, nbsizes, (n, n)) + ah.float().cpu().numpy()
                close_half(yh, ref, dtype)
    with pytest.raises(ValueError):
        hip.conv_gather_gemm(x, w, entry.fwd, addend=add[:-1])


def test_skip_gradient_rides_in_the_dgrad_write_back(hip, levels):
    """conv3d(..., with_skip=True): the input comes back routed through the convolution's autograd node; the gradient of
    f(conv(x)) + g(x) w.r.t. x equals autograd's sum of the two paths (fp32: to one addition's rounding; bf16 autocast: to the
    half rounding), and the convolution's own gradients are unchanged."""
    from openpcseg_amd import functional as F
    from openpcseg_amd.sparse import SparseTensor
    coords = t(levels[2])
    g = torch.Generator(device=DEV).manual_seed(5)
    w = (torch.randn(27, 64, 64, device=DEV, generator=g) / np.sqrt(64 * 27)).requires_grad_(True)
    x0 = torch.randn(coords.shape[0], 64, device=DEV, generator=g)
    gy = torch.randn(coords.shape[0], 64, device=DEV, generator=g)
    for amp in (None, torch.bfloat16):
        res = []
        for fused in (False, True):
            x = x0.clone().requires_grad_(True)
            w.grad = None
            with torch.autocast("cuda", dtype=amp, enabled=amp is not None):
                st = SparseTensor(x if amp is None else x.to(amp), coords, 2)
                if fused:
                    out, skip = F.conv3d(st, w, 3, with_skip=True)
                else:
                    out, skip = F.conv3d(st, w, 3), st
                y = torch.relu(out.feats.float()) * 2.0 + torch.tanh(skip.feats.float())
            y.backward(gy)
            res.append((x.grad.clone(), w.grad.clone()))
        (gx0, gw0), (gx1, gw1) = res
        tol = 1e-6 if amp is None else 2.0 ** -7
        assert float((gx1 - gx0).abs().max()) <= tol * float(gx0.abs().max())
        assert float((gw1 - gw0).abs().max()) <= 1e-6 * float(gw0.abs().max())


@pytest.mark.parametrize("amp", [None, torch.bfloat16])
def test_batchnorm_backward_statistics_from_the_dgrad_write_back(hip, levels, amp, monkeypatch):
    """conv -> BatchNorm -> ReLU -> conv (R:pcseg/model/segmentor/voxel/minkunet/minkunet.py:31-129): the second convolution's dgrad
    launch writes the BatchNorm's dy and leaves sum(g), sum(g xhat) per tile (pcs_conv_gather_gemm_*_ex, bn_x); the BatchNorm's
    backward reduces those instead of running pcs_bn_bwd_stats_*. Same gradients as the statistics pass; a second consumer of
    the activation (autograd then sums gradients) falls back to the pass."""
    from openpcseg_amd import fused, modules as spnn, native
    from openpcseg_amd.sparse import SparseTensor
    coords = t(levels[4])
    n = coords.shape[0]
    torch.manual_seed(3)
    c1, c2 = spnn.Conv3d(64, 96, 3).to(DEV), spnn.Conv3d(96, 64, 3).to(DEV)
    bn = fused.FusedBatchNorm(96).to(DEV).train()
    x0 = torch.randn(n, 64, device=DEV)
    gy = torch.randn(n, 64, device=DEV)
    calls = {"link": 0, "pass": 0}
    be = native.backend()
    real_link, real_pass = be.bn_bwd_reduce_partials, be.bn_bwd_stats
    monkeypatch.setattr(be, "bn_bwd_reduce_partials", lambda *a, **k: (calls.__setitem__("link", calls["link"] + 1), real_link(*a, **k))[1])
    monkeypatch.setattr(be, "bn_bwd_stats", lambda *a, **k: (calls.__setitem__("pass", calls["pass"] + 1), real_pass(*a, **k))[1])

    def run(link, second_consumer=False):
        monkeypatch.setattr(fused, "LINK_BN_BWD", link)
        for m in (c1, c2, bn):
            m.zero_grad(set_to_none=True)
        x = x0.clone().requires_grad_(True)
        with torch.autocast("cuda", dtype=amp, enabled=amp is not None):
            h = bn(c1(SparseTensor(x, coords, 4)), relu=True)
            out = c2(h).feats.float()
            if second_consumer:
                out = out + h.feats.float()[:, :64] * 0.5
        out.backward(gy)
        return [x.grad.clone()] + [p.grad.clone() for m in (c1, bn, c2) for p in m.parameters()]

    base = run(False)
    assert calls == {"link": 0, "pass": 1}
    got = run(True)
    assert calls == {"link": 1, "pass": 1}                      # the statistics came out of the dgrad write-back
    tol = 2e-5 if amp is None else 2e-3
    for a, b in zip(got, base):
        assert float((a - b).abs().max()) <= tol * max(float(b.abs().max()), 1e-6)
    both = run(True, second_consumer=True)
    assert calls == {"link": 1, "pass": 2}                      # two consumers of the activation: the pass runs
    ref = run(False, second_consumer=True)
    for a, b in zip(both, ref):
        assert float((a - b).abs().max()) <= tol * max(float(b.abs().max()), 1e-6)


def _ws_mode(hip, mode):
    import ctypes
    f = hip.lib.pcs_debug_convh_ws
    f.restype, f.argtypes = None, [ctypes.c_int32] * 3
    f(mode, 0, 0)


# every instance of the weight-stationary kernel (csrc/conv_wave6h.hip): (16-column tiles, 32-channel steps) incl. the thin ones its
# shape policy leaves to conv_os5h, the chunked contractions (192 ... 384 input channels, 4-wave tiles of <= 160 rows), two column
# tiles (cout 192 / 256), partial column tiles (cout 20 / 80), forced tile heights, 8-wave workgroups (tall tiles)
WS_CASES = [(1, 96, 96, None), (1, 96, 96, 384), (1, 128, 96, None), (1, 96, 128, 144), (4, 128, 128, None), (4, 128, 128, 288),
            (2, 64, 64, None), (8, 64, 128, 112), (4, 128, 64, None), (2, 64, 96, None), (2, 64, 32, None), (2, 32, 32, None),
            (4, 32, 64, None), (2, 32, 96, None), (4, 32, 128, 112), (8, 128, 256, 112), (4, 128, 192, 144), (8, 256, 256, 144),
            (8, 384, 256, 144), (4, 192, 128, 144), (8, 256, 128, 112), (4, 192, 96, 144), (8, 256, 96, 128),
            (4, 128, 80, None)]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("stride,cin,cout,tile", WS_CASES)
def test_weight_stationary_half_conv_dense_map(hip, levels, dtype, stride, cin, cout, tile):
    """conv_os6h_kernel (forced on for every shape it has an instance for) vs the oracle on the same half-rounded operands, the
    bound of the conv_os5h test; bit-reproducible; and against conv_os5h itself (same products, another fp32 addition order: they
    differ by output roundings only). Forward and dgrad (the input-sorted map with transposed weights)."""
    entry, nbmaps, nbsizes, n = level_map(levels, stride)
    rng = np.random.default_rng(stride * 100000 + cin * 100 + cout + 11)
    x = _round_half(rng.normal(size=(n, cin)).astype(np.float32), dtype)
    w = _round_half((rng.normal(size=(27, cin, cout)) / np.sqrt(cin * 27)).astype(np.float32), dtype)
    bias = rng.normal(size=cout).astype(np.float32)
    wp = hip.prepare_weights_h(t(w), dtype, transpose=False)
    dx = t(x).to(dtype)
    ref = orc.conv_fwd(x, w, nbmaps, nbsizes, (n, n))
    try:
        _ws_mode(hip, 0)
        y5 = hip.conv_gather_gemm_h(dx, wp, 27, cout, entry.fwd, tile_rows=tile)
        _ws_mode(hip, 3)
        y = hip.conv_gather_gemm_h(dx, wp, 27, cout, entry.fwd, tile_rows=tile)
        close_half(y, ref, dtype)
        assert torch.equal(y, hip.conv_gather_gemm_h(dx, wp, 27, cout, entry.fwd, tile_rows=tile))
        assert not torch.equal(y, y5) or cin <= 32            # it really ran another kernel (one-step layers may agree bit for bit)
        # two roundings of two fp32 sums that differ in their last bits: at most one unit of the storage format apart
        d, a5 = (y.float() - y5.float()).abs(), y5.float().abs()
        assert bool((d <= 2.0 * _HALF_TOL[dtype] * a5 + 8e-5 * float(a5.max())).all())
        close_half(hip.conv_gather_gemm_h(dx, wp, 27, cout, entry.fwd, bias=t(bias), tile_rows=tile), ref + bias[None, :], dtype)
        if cout % 8 == 0 and cout >= 32:                       # dgrad of the mirrored layer: cout -> cin over the input-sorted map
            gy = _round_half(rng.normal(size=(n, cout)).astype(np.float32), dtype)
            ogx, _ = orc.conv_bwd(x, gy, w, nbmaps, nbsizes)
            wpt = hip.prepare_weights_h(t(w), dtype, transpose=True)
            close_half(hip.conv_gather_gemm_h(t(gy).to(dtype), wpt, 27, cin, entry.rev, tile_rows=tile), ogx, dtype)
    finally:
        _ws_mode(hip, -1)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_weight_stationary_half_conv_bn_statistics(hip, levels, dtype):
    """The 16-byte-store epilogue of conv_os6h_kernel leaves the statistics of the ROUNDED values it stored (8 columns per thread,
    pivot-shifted sums un-shifted in double); outputs identical with and without them."""
    for stride, cin, cout in ((4, 128, 96), (4, 128, 128), (2, 64, 64)):
        entry, nbmaps, nbsizes, n = level_map(levels, stride)
        g = torch.Generator(device=DEV).manual_seed(3)
        x = (torch.randn(n, cin, device=DEV, generator=g) + 0.5).to(dtype)
        w = torch.randn(27, cin, cout, device=DEV, generator=g) / np.sqrt(cin * 27)
        wp = hip.prepare_weights_h(w, dtype, transpose=False)
        try:
            _ws_mode(hip, 3)
            got = []
            y = hip.conv_gather_gemm_h(x, wp, 27, cout, entry.fwd, bn_sums=got)
            assert torch.equal(y, hip.conv_gather_gemm_h(x, wp, 27, cout, entry.fwd)) and len(got) == 1
        finally:
            _ws_mode(hip, -1)
        yd = y.double()
        assert torch.allclose(got[0][:cout], yd.sum(0), rtol=0, atol=1e-6 * float(yd.abs().sum(0).max()))
        assert torch.allclose(got[0][cout:2 * cout], (yd * yd).sum(0), rtol=1e-6)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("stride,cin,cout", [(4, 128, 128), (8, 384, 256), (1, 128, 96), (2, 112, 56), (4, 336, 224),
                                             (8, 672, 448),
                                             # [r5] one-tile gradient blocks: the one-wave instance of wgrad3 (csrc/conv_wgrad.hip, THIN)
                                             (4, 64, 64), (2, 32, 32), (4, 32, 64), (4, 64, 32), (8, 64, 48)])
def test_half_conv_backward_dense_map(hip, levels, dtype, stride, cin, cout):
    """dgrad on the half kernel (weights re-packed with transpose=True) and the fp32-accumulated weight gradient from
    half operands (pcs_conv_wgrad_h) vs orc_conv_bwd on the half-rounded operands."""
    entry, nbmaps, nbsizes, n = level_map(levels, stride)
    rng = np.random.default_rng(stride * 100000 + cin * 100 + cout + 9)
    x = _round_half(rng.normal(size=(n, cin)).astype(np.float32), dtype)
    gy = _round_half(rng.normal(size=(n, cout)).astype(np.float32), dtype)
    w = _round_half((rng.normal(size=(27, cin, cout)) / np.sqrt(cin * 27)).astype(np.float32), dtype)
    ogx, ogw = orc.conv_bwd(x, gy, w, nbmaps, nbsizes)
    wpt = hip.prepare_weights_h(t(w), dtype, transpose=True)
    gx = hip.conv_gather_gemm_h(t(gy).to(dtype), wpt, 27, cin, entry.rev)
    close_half(gx, ogx, dtype)
    gw = hip.conv_wgrad_h(t(x).to(dtype), t(gy).to(dtype), entry.fwd, 0)
    assert gw.dtype == torch.float32
    close(gw, ogw, 2e-5)   # exact products of half operands, fp32 accumulation: the fp32 bound holds


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_conv3d_under_autocast(hip, levels, dtype):
    """conv3d under torch.autocast: >= 64-channel layers on the 16-bit MFMA kernels (half out), thin layers in fp32
    with a half-rounded output; gradients reach the fp32 master weights in fp32 and agree with the fp32 path to
    half precision."""
    from openpcseg_amd import functional as F
    from openpcseg_amd.sparse import SparseTensor
    c = t(levels[4])
    g = torch.Generator(device=DEV).manual_seed(11)
    x = torch.randn(c.shape[0], 64, device=DEV, generator=g)
    w1 = (torch.randn(27, 64, 128, device=DEV, generator=g) * 0.03).requires_grad_(True)
    w2 = (torch.randn(27, 128, 16, device=DEV, generator=g) * 0.03).requires_grad_(True)

    def run(amp):
        for w in (w1, w2):
            w.grad = None
        xs = SparseTensor(x.clone().requires_grad_(True), c, 4)
        with torch.autocast("cuda", dtype=dtype, enabled=amp):
            h = F.conv3d(xs, w1, 3)
            y = F.conv3d(h, w2, 3)
        y.F.float().square().sum().backward()
        return h.F, y.F, w1.grad.clone(), w2.grad.clone(), xs.F.grad.clone()

    h32, y32, g1, g2, gx = run(False)
    hh, yh, g1h, g2h, gxh = run(True)
    assert hh.dtype == dtype and yh.dtype == dtype and g1h.dtype == torch.float32 and gxh.dtype == torch.float32
    tol = 40 * _HALF_TOL[dtype]
    for a, b in ((hh, h32), (yh, y32), (g1h, g1), (g2h, g2), (gxh, gx)):
        assert (a.float() - b).abs().max() <= tol * b.abs().max(), ((a.float() - b).abs().max(), b.abs().max())


# ---- BatchNorm statistics from the convolution write-back ---------------------------------
@pytest.mark.parametrize("stride,cin,cout,tile", [(4, 128, 128, None), (8, 256, 256, None), (1, 96, 96, 384), (1, 32, 32, None),
                                                   (2, 32, 64, 128), (4, 128, 96, 112), (1, 4, 32, None)])
def test_conv_epilogue_bn_statistics(hip, levels, stride, cin, cout, tile):
    """[sum x | sum x^2 | n] of the conv output, produced per tile in the write-back and reduced in double, equals the
    float64 sums over the stored tensor (what pcs_bn_stats_f32 computes with one more read) -- and the output itself
    is bit-identical with and without the statistics."""
    entry, nbmaps, nbsizes, n = level_map(levels, stride)
    g = torch.Generator(device=DEV).manual_seed(cin + cout)
    x = torch.randn(n, cin, device=DEV, generator=g) + 0.5
    w = torch.randn(27, cin, cout, device=DEV, generator=g) / np.sqrt(cin * 27)
    got = []
    y = hip.conv_gather_gemm(x, w, entry.fwd, tile_rows=tile, bn_sums=got)
    assert torch.equal(y, hip.conv_gather_gemm(x, w, entry.fwd, tile_rows=tile))
    assert len(got) == 1 and got[0].shape == (2 * cout + 1,) and float(got[0][-1]) == n
    yd = y.double()
    assert torch.allclose(got[0][:cout], yd.sum(0), rtol=0, atol=1e-6 * float(yd.abs().sum(0).max()))
    assert torch.allclose(got[0][cout:2 * cout], (yd * yd).sum(0), rtol=1e-6)
    ref = hip.bn_stats(y)
    stat_a = hip.bn_finalize(got[0], float(n), 1e-5, 0.1, None, None)
    stat_b = hip.bn_finalize(ref, float(n), 1e-5, 0.1, None, None)
    assert torch.allclose(stat_a, stat_b, rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_half_conv_epilogue_bn_statistics(hip, levels, dtype):
    """Half kernel: the statistics are those of the ROUNDED values it stored."""
    entry, nbmaps, nbsizes, n = level_map(levels, 4)
    g = torch.Generator(device=DEV).manual_seed(3)
    x = (torch.randn(n, 128, device=DEV, generator=g) + 0.5).to(dtype)
    w = torch.randn(27, 128, 96, device=DEV, generator=g) / np.sqrt(128 * 27)
    wp = hip.prepare_weights_h(w, dtype, transpose=False)
    got = []
    y = hip.conv_gather_gemm_h(x, wp, 27, 96, entry.fwd, bn_sums=got)
    assert torch.equal(y, hip.conv_gather_gemm_h(x, wp, 27, 96, entry.fwd)) and len(got) == 1
    yd = y.double()
    assert torch.allclose(got[0][:96], yd.sum(0), rtol=0, atol=1e-6 * float(yd.abs().sum(0).max()))
    assert torch.allclose(got[0][96:192], (yd * yd).sum(0), rtol=1e-6)


def test_conv_x3_nonfinite_inputs_follow_the_fp32_kernel(hip, levels):
    """+-Inf features (and finite ones above the bf16 maximum, whose high plane rounds to Inf) make exactly the outputs
    they reach non-finite -- where the fp32 MFMA kernel gives Inf or, for the finite input, a huge finite value: the split
    keeps the lower planes of such a value zero (split3), so it poisons nothing else. Whether the poisoned output reads Inf or NaN is not preserved -- Inf times the (signed)
    lower planes of a weight is -Inf or NaN, inherent to a three-plane product -- and is documented so in DESIGN.md.
    NaN in, NaN out; every other output keeps the bf16x3 accuracy."""
    entry, nbmaps, nbsizes, n = level_map(levels, 4)
    rng = np.random.default_rng(11)
    x = rng.normal(size=(n, 64)).astype(np.float32)
    w = np.abs(rng.normal(size=(27, 64, 64)) / 40).astype(np.float32) + 1e-3   # positive weights: Inf * w stays +-Inf in fp32
    x[5, 3], x[900, 10], x[2000, 7] = np.inf, -np.inf, 3.4e38
    x[3000, 1] = np.nan
    dx, dw = t(x), t(w)
    y32 = hip.conv_gather_gemm(dx, dw, entry.fwd)
    y3 = hip.conv_gather_gemm_x3(dx, hip.prepare_weights_x3(dw, transpose=False), 27, 64, entry.fwd)
    # the outputs the four poisoned inputs reach: the same map on an indicator of those rows (dense positive weights: every column)
    ind = np.zeros_like(x)
    ind[[5, 900, 2000, 3000]] = 1.0
    touched = hip.conv_gather_gemm(t(ind), t(np.ones_like(w)), entry.fwd) > 0
    nan_touched = hip.conv_gather_gemm(t((ind * 0 + (np.arange(n) == 3000)[:, None]).astype(np.float32)), t(np.ones_like(w)), entry.fwd) > 0
    assert int(touched.sum()) > 0 and int(nan_touched.sum()) > 0
    assert torch.equal(~torch.isfinite(y3), touched)               # exactly the reached outputs, Inf or NaN
    assert bool((~torch.isfinite(y32) <= touched).all())           # the fp32 kernel: a subset (3.4e38 * w stays finite there)
    assert bool(torch.isnan(y3[nan_touched]).all()) and bool(torch.isnan(y32[nan_touched]).all())
    ok = ~touched
    assert float((y3[ok] - y32[ok]).abs().max()) <= 2e-5 * float(y32[ok].abs().max())


def test_commit_variants_bit_identical(hip, levels):
    """PCS_COMMIT_NOWAIT=0 / PCS_COMMIT_PHASED=0 (the fenced ticket hand-over and the compiler-interleaved commit kept behind
    macros in conv_wave5.hip, conv_wave5h.hip and conv_wave5x.hip) produce the same bits as the default build: a variant
    library is built here (hipcc is on the GPU box too) and both libraries run the same launches in subprocesses."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run(["bash", os.path.join(root, "tools", "build_variant_lib.sh"), "fenced", "-DPCS_COMMIT_NOWAIT=0",
                        "-DPCS_COMMIT_PHASED=0"], capture_output=True, text=True, timeout=900)
    lib = os.path.join(root, "openpcseg_amd", "lib", "dbg", "fenced.so")
    if r.returncode != 0 or not os.path.exists(lib):
        pytest.skip("variant library did not build here: " + r.stderr[-300:])
    code = r"""
import hashlib, sys, numpy as np, torch
sys.path.insert(0, %r)
from openpcseg_amd import functional as F, native
from openpcseg_amd.workloads.synthetic import make_batch
be = native.backend()
c = make_batch([0], n_points=30000)["lidar"].C.cuda()
c = c[torch.argsort(F.sphash(c))].contiguous()
e = F.build_kernel_map(c, c, (3, 3, 3), (1, 1, 1), (1, 1, 1))
torch.manual_seed(0)
h = hashlib.sha256()
for cin, cout in [(64, 64), (96, 96), (128, 96), (32, 32)]:
    x = torch.randn(c.shape[0], cin, device="cuda"); w = torch.randn(27, cin, cout, device="cuda") * 0.05
    h.update(be.conv_gather_gemm(x, w, e.fwd).cpu().numpy().tobytes())
    if cin >= 64:
        h.update(be.conv_gather_gemm_x3(x, be.prepare_weights_x3(w, transpose=False), 27, cout, e.fwd).cpu().numpy().tobytes())
        xb = x.bfloat16()
        h.update(be.conv_gather_gemm_h(xb, be.prepare_weights_h(w, torch.bfloat16, transpose=False), 27, cout, e.fwd).float().cpu().numpy().tobytes())
print("HASH", h.hexdigest())
""" % root
    out = []
    for extra in ({}, {"PCS_LIB_PATH": lib}):
        env = dict(os.environ, **extra)
        p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-800:]
        out.append([l for l in p.stdout.splitlines() if l.startswith("HASH")][-1])
    os.remove(lib)
    assert out[0] == out[1]


@pytest.mark.parametrize("amp", [None, torch.bfloat16])
@pytest.mark.parametrize("stride,cin,cout", [(4, 51, 102), (4, 102, 102), (2, 153, 51), (8, 409, 204), (8, 204, 409)])
def test_conv3d_odd_widths_run_padded_on_the_mfma_kernels(hip, levels, stride, cin, cout, amp):
    """The cr 1.6 model-zoo widths (R:tools/cfgs/voxel/waymo/minkunet_mk34_cr16.yaml: 51 / 102 / 153 / 204 / 409 channels) are not
    16-byte granular: conv3d zero-pads them onto the MFMA kernels (functional._channel_padding) instead of running the generic
    kernels. Forward, input gradient and weight gradient against the oracle on the unpadded operands; the launches are checked
    to be the padded shapes (and, under autocast, the half kernel)."""
    from openpcseg_amd import functional as F
    from openpcseg_amd.sparse import SparseTensor
    entry, nbmaps, nbsizes, n = level_map(levels, stride)
    rng = np.random.default_rng(stride * 1000 + cin + cout)
    x = rng.normal(size=(n, cin)).astype(np.float32)
    w = (rng.normal(size=(27, cin, cout)) / np.sqrt(cin * 27)).astype(np.float32)
    gy = rng.normal(size=(n, cout)).astype(np.float32)
    if amp is not None:
        x, w, gy = (_round_half(v, amp) for v in (x, w, gy))
    seen = []
    orig, orig_h = hip.conv_gather_gemm, hip.conv_gather_gemm_h
    hip.conv_gather_gemm = lambda src, weight, *a, **k: (seen.append(("f32", src.shape[1], weight.shape[2])), orig(src, weight, *a, **k))[1]
    hip.conv_gather_gemm_h = lambda src, wp, k_, co, *a, **k: (seen.append(("half", src.shape[1], co)), orig_h(src, wp, k_, co, *a, **k))[1]
    from openpcseg_amd import native
    prev = native._BACKEND
    native._BACKEND = hip
    try:
        xs = SparseTensor(t(x).requires_grad_(True), t(levels[stride]), stride)
        wt = t(w).requires_grad_(True)
        with torch.autocast("cuda", dtype=amp or torch.bfloat16, enabled=amp is not None):
            y = F.conv3d(xs, wt, 3)
        assert y.F.shape == (n, cout)
        y.F.backward(t(gy).to(y.F.dtype))
    finally:
        hip.conv_gather_gemm, hip.conv_gather_gemm_h, native._BACKEND = orig, orig_h, prev
    q = 8 if amp is not None else 4
    pc = lambda c: (c + q - 1) // q * q
    assert ("half" if amp is not None else "f32", pc(cin), pc(cout)) in seen and not any(s[1] % q or s[2] % q for s in seen), seen
    ref = orc.conv_fwd(x, w, nbmaps, nbsizes, (n, n))
    ogx, ogw = orc.conv_bwd(x, gy, w, nbmaps, nbsizes)
    if amp is None:
        close(y.F, ref, 2e-5)
        close(xs.F.grad, ogx, 2e-5)
        close(wt.grad, ogw, 2e-5)
    else:
        close_half(y.F, ref, amp)
        close_half(xs.F.grad.to(amp), ogx, amp)
        close(wt.grad, ogw, 2e-5)


@pytest.mark.parametrize("amp", [None, torch.bfloat16])
def test_pointwise_conv_odd_widths(hip, amp):
    """1x1x1 convolutions of the cr 1.6 widths (204 -> 153 ...): the weight gradient x^T dy runs zero-padded on the MFMA
    weight-gradient kernels; forward / input gradient are dense GEMMs. Against float64 matrix products."""
    from openpcseg_amd import functional as F
    from openpcseg_amd.sparse import SparseTensor
    rng = np.random.default_rng(3)
    n, cin, cout = 50000, 204, 153
    c = np.unique(np.concatenate([rng.integers(0, 80, size=(n, 3)), np.zeros((n, 1), np.int64)], 1).astype(np.int32), axis=0)
    n = c.shape[0]
    x = rng.normal(size=(n, cin)).astype(np.float32)
    w = (rng.normal(size=(cin, cout)) / np.sqrt(cin)).astype(np.float32)
    gy = rng.normal(size=(n, cout)).astype(np.float32)
    seen = []
    orig, orig_h = hip.conv_wgrad, hip.conv_wgrad_h
    hip.conv_wgrad = lambda fa, fb, *a, **k: (seen.append((fa.shape[1], fb.shape[1])), orig(fa, fb, *a, **k))[1]
    hip.conv_wgrad_h = lambda fa, fb, *a, **k: (seen.append((fa.shape[1], fb.shape[1])), orig_h(fa, fb, *a, **k))[1]
    from openpcseg_amd import native
    prev = native._BACKEND
    native._BACKEND = hip
    try:
        xs = SparseTensor(t(x).requires_grad_(True), t(c))
        wt = t(w).requires_grad_(True)
        with torch.autocast("cuda", dtype=amp or torch.bfloat16, enabled=amp is not None):
            y = F.conv3d(xs, wt, 1)
        y.F.backward(t(gy).to(y.F.dtype))
    finally:
        hip.conv_wgrad, hip.conv_wgrad_h, native._BACKEND = orig, orig_h, prev
    assert seen == [(204, 156)], seen
    tol = 2e-5 if amp is None else 2e-2
    x64, w64, g64 = x.astype(np.float64), w.astype(np.float64), gy.astype(np.float64)
    for got, ref in ((y.F, x64 @ w64), (xs.F.grad, g64 @ w64.T), (wt.grad, x64.T @ g64)):
        ref_t = torch.from_numpy(ref)
        assert float((got.double().cpu() - ref_t).abs().max()) <= tol * float(ref_t.abs().max())
