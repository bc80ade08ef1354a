"""Parity of the HIP path (through the C ABI) with the oracle and the reference goldens.
Run with `-m gpu` on an MI355X. Integer / index work: bit-exact. fp32: tolerance stated per test.
"""
import os
import numpy as np
import pytest
import torch

from oracle import oracle as orc

pytestmark = pytest.mark.gpu

DEV = "cuda"


def t(a, dtype=None):
    x = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        x = x.to(dtype)
    return x.to(DEV)


def close(a, b, rtol=1e-4):
    """fp32 MFMA / FMA vs scalar-C summation order: relative to the tensor's max magnitude."""
    a = a.detach().cpu().numpy().astype(np.float64) if isinstance(a, torch.Tensor) else np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    scale = max(np.abs(b).max(), 1e-6) if b.size else 1.0
    err = np.abs(a - b).max() / scale if b.size else 0.0
    assert err <= rtol, err


def random_scene(rng, n, extent, batches):
    c = np.concatenate([rng.integers(0, extent, size=(n, 3)), rng.integers(0, batches, size=(n, 1))], axis=1)
    c = np.unique(c.astype(np.int32), axis=0)
    return c[rng.permutation(c.shape[0])]


# ---- K1 / K2 -------------------------------------------------------------------------------------
def test_hash_golden_and_random(hip, golden):
    assert (hip.hash(t(golden["kat_coords"])).cpu().numpy() == golden["kat_hash"]).all()
    assert (hip.hash(t(golden["hash_coords"])).cpu().numpy() == golden["hash_out"]).all()
    kh = hip.kernel_hash(t(golden["hash_coords"]), t(golden["khash_offsets"]))
    assert (kh.cpu().numpy() == golden["khash_out"]).all()
    rng = np.random.default_rng(0)
    c = rng.integers(-2 ** 31, 2 ** 31 - 1, size=(100003, 4)).astype(np.int32)
    assert (hip.hash(t(c)).cpu().numpy() == orc.sphash(c)).all()
    off = orc.get_kernel_offsets((3, 1, 3), 4)
    assert (hip.kernel_hash(t(c[:5000]), t(off)).cpu().numpy() == orc.sphash(c[:5000], off)).all()
    assert hip.hash(torch.zeros(0, 4, dtype=torch.int32, device=DEV)).shape == (0,)


# ---- K3-K5 / K6 ----------------------------------------------------------------------------------
def test_hash_query(hip, golden):
    ref_h = orc.sphash(golden["scene_coords"])
    out = hip.hash_query(t(golden["query_q"]), t(ref_h))
    assert (out.cpu().numpy() == golden["query_out"]).all()
    # duplicates: smallest position wins; misses -> -1; empty query / empty table
    out = hip.hash_query(t(np.array([5, 7, 9], np.int64)), t(np.array([7, 5, 7, 5], np.int64)))
    assert out.cpu().tolist() == [1, 0, -1]
    assert hip.hash_query(t(np.zeros(0, np.int64)), t(ref_h)).numel() == 0
    assert hip.hash_query(t(np.array([3], np.int64)), t(np.zeros(0, np.int64))).cpu().tolist() == [-1]
    rng = np.random.default_rng(1)
    keys = rng.integers(0, 2 ** 60, size=300000)
    q = np.concatenate([keys[::2], rng.integers(0, 2 ** 60, size=1000)])
    assert (hip.hash_query(t(q), t(keys)).cpu().numpy() == orc.sphashquery(q, keys)).all()


def test_count(hip):
    rng = np.random.default_rng(2)
    idx = rng.integers(-1, 5000, size=200000).astype(np.int32)
    assert (hip.count(t(idx), 5000).cpu().numpy() == orc.spcount(idx, 5000)).all()
    assert hip.count(t(np.zeros(0, np.int32)), 7).cpu().tolist() == [0] * 7


# ---- K7-K10 ----------------------------------------------------------------------------------------
@pytest.mark.parametrize("c", [3, 4, 6, 32, 96])
def test_voxelize(hip, c):
    rng = np.random.default_rng(c)
    n, m = 50000, 9000
    idx = rng.integers(0, m, size=n).astype(np.int32)
    idx[:17] = -1
    counts = orc.spcount(idx, m)
    feats = rng.normal(size=(n, c)).astype(np.float32)
    want = orc.voxelize_fwd(feats, idx, counts)
    di = t(idx)
    out = hip.voxelize_fwd(t(feats), di, t(counts))  # segmented (sorted points), order cached on the index tensor
    close(out, want, 1e-5)
    assert di._pcs_vox_csr[0][-2] == m and torch.equal(out, hip.voxelize_fwd(t(feats), di, t(counts)))  # deterministic
    close(hip.voxelize_fwd_atomic(t(feats), t(idx), t(counts)), want, 1e-5)  # the reference's atomic dataflow
    gout = rng.normal(size=(m, c)).astype(np.float32)
    close(hip.voxelize_bwd(t(gout), t(idx), t(counts), n), orc.voxelize_bwd(gout, idx, counts, n), 1e-6)


def test_voxelize_golden(hip, golden):
    out = hip.voxelize_fwd(t(golden["vox_feats"]), t(golden["vox_idx"]), t(golden["vox_counts"]))
    close(out, golden["vox_out"], 1e-6)


@pytest.mark.parametrize("c,m", [(5, 4000), (32, 4000), (96, 4000), (256, 4000), (20, 4000), (20, 37), (8, 100000)])
def test_devoxelize(hip, c, m):
    """c <= 32 (16-byte granular) takes the wave-per-voxel backward kernel: m = 37 -> ~6 500 entries per voxel, m = 100 000 ->
    mostly empty and one-entry voxels."""
    rng = np.random.default_rng(c + m)
    n = 30000
    idx8 = rng.integers(-1, m, size=(n, 8)).astype(np.int32)
    w8 = rng.uniform(0, 1, size=(n, 8)).astype(np.float32)
    feat = rng.normal(size=(m, c)).astype(np.float32)
    close(hip.devoxelize_fwd(t(feat), t(idx8), t(w8)), orc.devoxelize_fwd(feat, idx8, w8), 1e-6)
    gout = rng.normal(size=(n, c)).astype(np.float32)
    close(hip.devoxelize_bwd(t(gout), t(idx8), t(w8), m), orc.devoxelize_bwd(gout, idx8, w8, m), 1e-5)


def test_ti_weights_golden(hip, golden):
    for s in (1, 2, 4):
        w = hip.ti_weights(t(golden["tiw_coords"]), t(golden["tiw_idxq"]), s)
        close(w, golden["tiw_s%d" % s], 2e-6)
    out = hip.devoxelize_fwd(t(golden["devox_feat"]), t(np.ascontiguousarray(golden["tiw_idxq"].T), torch.int32),
                             t(np.ascontiguousarray(golden["tiw_s2"].T)))
    close(out, golden["devox_out"], 1e-6)


# ---- downsample + rulebook: bit-exact incl. order -------------------------------------------------
@pytest.mark.parametrize("name,args", [("k2s2", (2, 2, 1)), ("k3s2", (2, 3, 1)), ("k3s221", ((2, 2, 1), 3, 1))])
def test_downsample_golden(hip, golden, name, args):
    from openpcseg_amd import functional as F
    out = F.spdownsample(t(golden["scene_coords"]), *args)
    assert out.dtype == torch.int32 and (out.cpu().numpy() == golden["ds_" + name]).all()


def test_downsample_random(hip):
    from openpcseg_amd import functional as F
    rng = np.random.default_rng(3)
    c = random_scene(rng, 60000, 90, 4)
    for args in [(2, 2, 1), (2, 2, 4), ((2, 2, 1), 3, 2), (2, 3, 1)]:
        cc = c.copy()
        ts = args[2]
        cc[:, :3] *= ts
        assert (F.spdownsample(t(cc), *args).cpu().numpy() == orc.spdownsample(cc, *args)).all()


@pytest.mark.parametrize("n,m", [(0, 5), (1, 1), (1000, 1), (4096, 65535), (4097, 65536), (300000, 36068), (3000000, 1158864)])
def test_index_csr(hip, n, m):
    """pcs_index_csr_i32: entries sorted by target row, equal rows in ascending position (what a stable sort gives), entries
    without a row (negative or >= m) outside [rowptr[0], rowptr[m]); row counts at the power-of-two edges of the key width."""
    rng = np.random.default_rng(n + m)
    idx = rng.integers(-1, m, size=n, dtype=np.int64).astype(np.int32)
    if n > 10:
        idx[rng.integers(0, n, size=5)] = m + 3          # beyond the last row: no row
        idx[rng.integers(0, n, size=5)] = m - 1          # the last row is used
    order, rowptr = hip._csr(t(idx), m)
    order, rowptr = order.cpu().numpy(), rowptr.cpu().numpy()
    assert rowptr.shape == (m + 1,) and order.shape == (n,) and np.array_equal(np.sort(order), np.arange(n))
    valid = (idx >= 0) & (idx < m)
    counts = np.bincount(idx[valid], minlength=m)
    assert rowptr[0] == 0 and np.array_equal(np.diff(rowptr), counts) and rowptr[m] == valid.sum()
    ref = np.argsort(np.where(valid, idx, m), kind="stable")
    assert np.array_equal(order[:rowptr[m]], ref[:rowptr[m]])


def test_prebuild_coords_are_the_lazy_ones(hip):
    """functional.prebuild_coords leaves in cmaps exactly what the strided convolutions would compute themselves (both
    spdownsample branches: k2 s2 fast, k3 s2 general), so a network runs bit-identically with its levels built up front."""
    from openpcseg_amd import functional as F
    from openpcseg_amd.sparse import SparseTensor
    rng = np.random.default_rng(11)
    c = random_scene(rng, 30000, 70, 3)
    feats = rng.normal(size=(c.shape[0], 8)).astype(np.float32)
    ws = [t((rng.normal(size=(8, 8, 8)) * 0.1).astype(np.float32)), t((rng.normal(size=(27, 8, 8)) * 0.1).astype(np.float32)),
          t((rng.normal(size=(8, 8, 8)) * 0.1).astype(np.float32))]
    steps = [(2, 2), (2, 3), (2, 2)]

    def run(prebuild):
        x = SparseTensor(t(feats), t(c))
        if prebuild:
            F.prebuild_coords(x, steps)
            assert sorted(x.cmaps) == [(2, 2, 2), (4, 4, 4), (8, 8, 8)]
        for w, (s, k) in zip(ws, steps):
            x = F.conv3d(x, w, k, stride=s)
        return x

    a, b = run(False), run(True)
    assert sorted(a.cmaps) == sorted(b.cmaps) and all(torch.equal(a.cmaps[k], b.cmaps[k]) for k in a.cmaps)
    assert torch.equal(a.C, b.C) and torch.equal(a.F, b.F)


@pytest.mark.parametrize("n", [0, 1, 2, 63, 64, 65, 4097, 300000, 2500000])
def test_sort_unique_keys(hip, n):
    """pcs_sort_unique_i64 (step 2 of spdownsample, TS:torchsparse/nn/functional/downsample.py:47-51): the distinct keys in
    ascending SIGNED order = numpy's unique, the one-record read-back (count, largest key, error flag), duplicates of every
    multiplicity, negative keys, the INT64_MAX sentinel, empty and one-element inputs; the size query is what the call checks."""
    rng = np.random.default_rng(n + 5)
    keys = rng.integers(-(1 << 62), 1 << 62, size=n, dtype=np.int64)
    if n > 2:
        keys[rng.integers(0, n, size=n // 2)] = keys[rng.integers(0, n, size=n // 2)]   # duplicates
        keys[rng.integers(0, n, size=max(n // 50, 1))] = np.iinfo(np.int64).max          # the sentinel, many times
        keys[rng.integers(0, n, size=max(n // 70, 1))] = rng.integers(-5, 5, size=max(n // 70, 1))
    ref = np.unique(keys)
    err = torch.full((1,), 7, dtype=torch.int32, device=DEV)
    buf, info = hip.sort_unique(t(keys), err)
    m, last, flag = info.tolist()
    assert m == ref.size and flag == 7
    assert last == (int(ref[-1]) if ref.size else np.iinfo(np.int64).min)
    assert (buf[:m].cpu().numpy() == ref).all()
    buf2, info2 = hip.sort_unique(t(keys))
    assert info2.tolist() == [m, last, 0] and torch.equal(buf2[:m], buf[:m])
    if n > 0:
        ws = torch.empty(64, dtype=torch.uint8, device=DEV)
        rc = hip.lib.pcs_sort_unique_i64(t(keys).data_ptr(), n, buf.data_ptr(), info.data_ptr(), None, ws.data_ptr(), 64, None)
        assert rc != 0 and hip.lib.pcs_sort_unique_ws_bytes(n) > 64 and b"workspace" in hip.lib.pcs_last_error()


@pytest.mark.parametrize("name,ks,in_stride", [("k3s1", 3, 1), ("k2s2", 2, 1), ("k133", (1, 3, 3), 1),
                                                ("k313", (3, 1, 3), 1), ("k3s2", 3, 1), ("k3s221", 3, 1)])
def test_kmap_golden(hip, golden, name, ks, in_stride):
    from openpcseg_amd import functional as F
    inc = golden["scene_coords"]
    outc = golden["ds_" + name] if ("ds_" + name) in golden.files else inc
    entry = F.build_kernel_map(t(inc), t(outc), (ks,) * 3 if isinstance(ks, int) else ks, (in_stride,) * 3, (1, 1, 1))
    assert (entry[1].cpu().numpy() == golden["kmap_%s_nbsizes" % name]).all()
    assert (entry[0].cpu().numpy().astype(np.int64) == golden["kmap_%s_nbmaps" % name]).all()
    assert entry[2] == (inc.shape[0], outc.shape[0])
    # the input-sorted map holds the same pair set, sorted by input row within each offset
    fwd, rev = entry.fwd, entry.rev
    assert rev.koff_host == fwd.koff_host
    fp, rp = fwd.pairs.cpu().numpy(), rev.pairs.cpu().numpy()
    for k in range(fwd.K):
        a, b = fwd.koff_host[k], fwd.koff_host[k + 1]
        f = fp[a:b]
        r = rp[a:b]
        assert (np.diff(r[:, 1]) > 0).all()
        order = np.argsort(f[:, 0], kind="stable")
        assert (f[order][:, [1, 0]] == r).all()


def test_kmap_random_multibatch(hip):
    from openpcseg_amd import functional as F
    rng = np.random.default_rng(4)
    c = random_scene(rng, 40000, 60, 3)
    for ks, st in [(3, 1), ((3, 1, 1), 1), (2, 2), (3, (2, 2, 1))]:
        ks3 = (ks,) * 3 if isinstance(ks, int) else ks
        st3 = (st,) * 3 if isinstance(st, int) else st
        outc = c if all(s == 1 for s in st3) else orc.spdownsample(c, st3, ks3, 1)
        nbmaps, nbsizes = orc.build_kmap(c, outc, ks3, 1)
        entry = F.build_kernel_map(t(c), t(outc), ks3, (1, 1, 1), (1, 1, 1))
        assert (entry[1].cpu().numpy() == nbsizes).all()
        assert (entry[0].cpu().numpy().astype(np.int64) == nbmaps).all()


# ---- convolution -----------------------------------------------------------------------------------
def _scene_maps(hip, golden, which):
    from openpcseg_amd import functional as F
    inc = golden["scene_coords"]
    if which == "k3s1":
        return F.build_kernel_map(t(inc), t(inc), (3, 3, 3), (1, 1, 1), (1, 1, 1)), \
            (golden["kmap_k3s1_nbmaps"], golden["kmap_k3s1_nbsizes"]), inc.shape[0], inc.shape[0]
    outc = golden["ds_k2s2"]
    return F.build_kernel_map(t(inc), t(outc), (2, 2, 2), (1, 1, 1), (1, 1, 1)), \
        (golden["kmap_k2s2_nbmaps"], golden["kmap_k2s2_nbsizes"]), inc.shape[0], outc.shape[0]


@pytest.mark.parametrize("n", [1, 17, 40])
def test_conv_on_tiny_maps_with_tile_order(hip, n):
    """A handful of voxels: one (partly filled) row tile, a tile order of one entry, most offsets empty -- every wave
    kernel (fp32 32 / 64 / 128 / 256 columns, half) against the oracle."""
    from openpcseg_amd import functional as F
    rng = np.random.default_rng(n)
    cube = np.stack(np.meshgrid(np.arange(4), np.arange(4), np.arange(3), indexing="ij"), -1).reshape(-1, 3)
    c = np.concatenate([cube[rng.permutation(len(cube))[:n]], np.zeros((n, 1), np.int64)], 1).astype(np.int32)
    entry = F.build_kernel_map(t(c), t(c), (3, 3, 3), (1, 1, 1), (1, 1, 1))
    nbmaps, nbsizes = orc.build_kmap(c, c, 3)
    assert np.array_equal(entry[0].cpu().numpy().astype(np.int64), nbmaps)
    assert hip._tile_order(entry.fwd, 128).cpu().tolist() == [0]
    for cin, cout in ((32, 32), (64, 64), (96, 128), (128, 256)):
        x = rng.normal(size=(n, cin)).astype(np.float32)
        w = (rng.normal(size=(27, cin, cout)) / np.sqrt(cin * 27)).astype(np.float32)
        ref = orc.conv_fwd(x, w, nbmaps, nbsizes, (n, n))
        close(hip.conv_gather_gemm(t(x), t(w), entry.fwd, ordered="force"), ref, 2e-5)
        close(hip.conv_gather_gemm(t(x), t(w), entry.fwd, tile_rows=256), ref, 2e-5)
        xh, wp = t(x).bfloat16(), hip.prepare_weights_h(t(w), torch.bfloat16, transpose=False)
        refh = orc.conv_fwd(xh.float().cpu().numpy(), t(w).bfloat16().float().cpu().numpy(), nbmaps, nbsizes, (n, n))
        close(hip.conv_gather_gemm_h(xh, wp, 27, cout, entry.fwd, ordered="force").float(), refh, 1e-2)


@pytest.mark.parametrize("cin,cout", [(4, 32), (32, 32), (32, 64), (96, 96), (128, 96), (192, 128), (256, 256),
                                      (384, 256), (64, 20), (5, 33), (56, 112)])
@pytest.mark.parametrize("tile", [64, 128])
def test_conv_forward_and_dgrad_kernel(hip, golden, cin, cout, tile):
    entry, (nbmaps, nbsizes), n_in, n_out = _scene_maps(hip, golden, "k3s1")
    rng = np.random.default_rng(cin * 1000 + cout)
    x = rng.normal(size=(n_in, cin)).astype(np.float32)
    w = (rng.normal(size=(27, cin, cout)) / np.sqrt(cin * 27)).astype(np.float32)
    y = hip.conv_gather_gemm(t(x), t(w), entry.fwd, tile_rows=tile)
    close(y, orc.conv_fwd(x, w, nbmaps, nbsizes, (n_in, n_out)), 2e-5)
    # dgrad = the same kernel on the input-sorted map with per-offset transposed weights
    gy = rng.normal(size=(n_out, cout)).astype(np.float32)
    gx = hip.conv_gather_gemm(t(gy), t(np.ascontiguousarray(w.transpose(0, 2, 1))), entry.rev, tile_rows=tile)
    ogx, ogw = orc.conv_bwd(x, gy, w, nbmaps, nbsizes)
    close(gx, ogx, 2e-5)
    close(hip.conv_wgrad(t(x), t(gy), entry.fwd, 0), ogw, 2e-5)


@pytest.mark.parametrize("cin,cout", [(96, 96), (128, 256), (64, 64)])
@pytest.mark.parametrize("tile", [16, 80, 112, 144, 160, 224, 256, 288, 384])
def test_conv_free_tile_heights(hip, golden, cin, cout, tile):
    """cin >= 64 kernel: any multiple of 16 is a legal tile height (the per-layer pick uses 80..160). The commit order
    (full row-block groups, then partial ones) depends on the tile, so heights agree to rounding, and one height is
    bit-reproducible run to run."""
    entry, (nbmaps, nbsizes), n_in, n_out = _scene_maps(hip, golden, "k3s1")
    rng = np.random.default_rng(cin + cout)
    x = rng.normal(size=(n_in, cin)).astype(np.float32)
    w = (rng.normal(size=(27, cin, cout)) / np.sqrt(cin * 27)).astype(np.float32)
    y = hip.conv_gather_gemm(t(x), t(w), entry.fwd, tile_rows=tile)
    close(y, orc.conv_fwd(x, w, nbmaps, nbsizes, (n_in, n_out)), 2e-5)
    close(y, hip.conv_gather_gemm(t(x), t(w), entry.fwd, tile_rows=128).cpu().numpy(), 1e-6)
    assert torch.equal(y, hip.conv_gather_gemm(t(x), t(w), entry.fwd, tile_rows=tile))


def test_submanifold_reverse_map_mirror_equals_probe(hip, golden):
    from openpcseg_amd import functional as F
    c = t(golden["scene_coords"])
    entry = F.build_kernel_map(c, c, (3, 3, 3), (1, 1, 1), (1, 1, 1))
    assert entry._mirror
    built = hip.build_kmap(c, c, -entry._ctx[2])
    assert torch.equal(entry.rev.pairs, built.pairs) and torch.equal(entry.rev.koff, built.koff)
    assert entry.rev.koff_host == built.koff_host and torch.equal(entry.rev.nbsizes, built.nbsizes)
    assert (entry.rev.n_src, entry.rev.n_dst) == (built.n_src, built.n_dst)


@pytest.mark.parametrize("shape", [(27, 256, 128), (8, 96, 32), (1, 20, 480), (27, 5, 33)])
def test_weight_transpose(hip, shape):
    w = torch.randn(*shape, device=DEV)
    assert torch.equal(hip.transpose_weights(w), w.transpose(1, 2).contiguous())


def test_rulebook_sizes_are_deferred(hip, golden):
    """build_kmap returns without waiting for the per-offset sizes: the forward launches (segment table, fused conv)
    run off the device-side offsets; koff_host / pairs / num_pairs resolve on first read and match the reference."""
    from openpcseg_amd import functional as F
    inc = golden["scene_coords"]
    entry = F.build_kernel_map(t(inc), t(inc), (3, 3, 3), (1, 1, 1), (1, 1, 1))
    km = entry.fwd
    assert not km.resolved and km._pairs_raw.shape[0] == 27 * inc.shape[0]  # worst-case buffer, sizes still in flight
    x = torch.randn(inc.shape[0], 64, device=DEV)
    w = torch.randn(27, 64, 64, device=DEV) * 0.05
    y = hip.conv_gather_gemm(x, w, km)  # launch shape from the estimate, no host read
    assert not km.resolved
    nbmaps, nbsizes = golden["kmap_k3s1_nbmaps"], golden["kmap_k3s1_nbsizes"]
    assert km.num_pairs == int(nbsizes.sum()) and km.resolved
    assert np.array_equal(entry[0].cpu().numpy(), nbmaps) and km._pairs_raw.shape[0] == km.num_pairs
    nb0, nb1, sizes = entry  # unpacks like the reference's kmap entry
    assert nb0 is entry[0] and sizes == (inc.shape[0], inc.shape[0])
    close(y, orc.conv_fwd(x.cpu().numpy(), w.cpu().numpy(), nbmaps, nbsizes, (inc.shape[0],) * 2), 2e-5)
    assert km.num_pairs_estimate() == km.num_pairs


def test_conv_tile_pick_and_errors(hip, golden):
    lib = hip.lib
    # the picks behind profiles/round2_tile_sweep_{f32,bf16}.md (12-frame bench maps): 96 columns on the sparse levels:
    # the tallest tile (one 8-wave workgroup per CU); <= 64-column tiles (64 outputs, or >= 128 outputs split in
    # 64-column tiles): 192..288 rows by the per-CU cost model (36k rows x 4 column tiles at 288 rows = 504 workgroups,
    # two per CU); half kernels: up to the tallest tile that fits twice per CU
    assert lib.pcs_conv_pick_tile_rows(1158864, 5112372, 27, 96, 96) == 384
    assert lib.pcs_conv_pick_tile_rows(329421, 2752033, 27, 128, 128) == 288
    assert lib.pcs_conv_pick_tile_rows(36068, 331722, 27, 256, 256) == 288
    assert lib.pcs_conv_pick_tile_rows(113008, 1001308, 27, 128, 128) == 224
    assert lib.pcs_conv_pick_tile_rows(113008, 1001308, 27, 256, 256) == 256
    assert lib.pcs_conv_pick_tile_rows(688382, 3790760, 27, 32, 32) in (224, 256, 288)  # 32-channel layers: the same kernel
    assert lib.pcs_conv_pick_tile_rows(36068, 331722, 27, 16, 32) == 128  # cin % 32 != 0: not the row-block-group kernel
    assert lib.pcs_conv_pick_tile_rows_dt(1158864, 5112372, 27, 96, 96, 0) == 384
    assert lib.pcs_conv_pick_tile_rows_dt(1158864, 5112372, 27, 96, 96, 1) == 192
    assert lib.pcs_conv_pick_tile_rows_dt(329421, 2752033, 27, 128, 128, 1) == 144
    assert lib.pcs_conv_pick_tile_rows_dt(113008, 1001308, 27, 128, 128, 2) == 112
    assert lib.pcs_conv_pick_tile_rows_dt(36068, 331722, 27, 256, 256, 1) == 144   # [r6] the small stride-16 level: the weight-stationary kernel on two 4-wave tiles per CU
    entry, _, n_in, _ = _scene_maps(hip, golden, "k3s1")
    x = torch.zeros((n_in, 16), device="cuda")
    w = torch.zeros((27, 16, 32), device="cuda")
    for bad in (8, 24, 1024, 144):  # 144 is legal only for the row-block-group kernel (cin % 32 == 0)
        with pytest.raises(RuntimeError):
            hip.conv_gather_gemm(x, w, entry.fwd, tile_rows=bad)


@pytest.mark.parametrize("cin,cout", [(32, 32), (64, 128), (256, 128), (96, 96)])
def test_conv_strided_and_transposed(hip, golden, cin, cout):
    entry, (nbmaps, nbsizes), n_in, n_out = _scene_maps(hip, golden, "k2s2")
    rng = np.random.default_rng(cin + cout)
    w = (rng.normal(size=(8, cin, cout)) / np.sqrt(cin * 8)).astype(np.float32)
    x = rng.normal(size=(n_in, cin)).astype(np.float32)
    close(hip.conv_gather_gemm(t(x), t(w), entry.fwd), orc.conv_fwd(x, w, nbmaps, nbsizes, (n_in, n_out)), 2e-5)
    xc = rng.normal(size=(n_out, cin)).astype(np.float32)  # transposed: input lives on the coarse rows
    close(hip.conv_gather_gemm(t(xc), t(w), entry.rev),
          orc.conv_fwd(xc, w, nbmaps, nbsizes, (n_in, n_out), transposed=True), 2e-5)
    gy = rng.normal(size=(n_in, cout)).astype(np.float32)
    ogx, ogw = orc.conv_bwd(xc, gy, w, nbmaps, nbsizes, transposed=True)
    close(hip.conv_gather_gemm(t(gy), t(np.ascontiguousarray(w.transpose(0, 2, 1))), entry.fwd), ogx, 2e-5)
    close(hip.conv_wgrad(t(xc), t(gy), entry.fwd, 1), ogw, 2e-5)


@pytest.mark.parametrize("tag,ks,stride,transposed", [("conv_k3s1_N", 3, 1, False), ("conv_k2s2_N", 2, 2, False),
                                                       ("conv_k2s2_T", 2, 2, True)])
def test_conv3d_autograd_vs_reference_golden(hip, golden, tag, ks, stride, transposed):
    from openpcseg_amd import functional as F
    from openpcseg_amd.sparse import SparseTensor
    coords = t(golden["scene_coords"])
    x = t(golden[tag + "_x"]).requires_grad_(True)
    w = t(golden[tag + "_w"]).requires_grad_(True)
    if not transposed:
        out = F.conv3d(SparseTensor(x, coords, 1), w, ks, stride=stride)
    else:
        fine = SparseTensor(torch.zeros(coords.shape[0], 8, device=DEV), coords, 1)
        fine.cmaps[(1, 1, 1)] = coords
        down = F.conv3d(fine, torch.zeros(8, 8, 12, device=DEV), ks, stride=stride)
        inp = SparseTensor(x, down.C, down.s)
        inp.cmaps, inp.kmaps = down.cmaps, down.kmaps
        out = F.conv3d(inp, w, ks, stride=stride, transposed=True)
    close(out.F, golden[tag + "_y"], 2e-5)
    out.F.backward(t(golden[tag + "_gy"]))
    close(x.grad, golden[tag + "_gx"], 2e-5)
    close(w.grad, golden[tag + "_gw"], 2e-5)


def test_conv_empty_and_ragged(hip):
    from openpcseg_amd import functional as F
    # isolated voxels: only the centre offset has pairs; 1 voxel; 129 voxels (tile boundary + 1)
    for n in (1, 129, 130):
        c = np.zeros((n, 4), np.int32)
        c[:, 0] = np.arange(n) * 5
        entry = F.build_kernel_map(t(c), t(c), (3, 3, 3), (1, 1, 1), (1, 1, 1))
        assert entry[1].cpu().tolist() == [0] * 13 + [n] + [0] * 13
        x = np.random.default_rng(n).normal(size=(n, 32)).astype(np.float32)
        w = np.random.default_rng(n + 1).normal(size=(27, 32, 32)).astype(np.float32)
        close(hip.conv_gather_gemm(t(x), t(w), entry.fwd), x @ w[13], 2e-5)
        close(hip.conv_wgrad(t(x), t(x), entry.fwd, 0)[13], x.T @ x, 2e-5)
        assert float(hip.conv_wgrad(t(x), t(x), entry.fwd, 0)[0].abs().max()) == 0.0


# ---- end to end ------------------------------------------------------------------------------------
def test_minkunet_logits_match_reference(hip, golden_e2e):
    """per-point logits within 1e-3 (fp32) of the reference's MinkUNet + reference backend."""
    from seeded import seeded_state
    from openpcseg_amd.sparse import SparseTensor
    from openpcseg_amd.workloads.minkunet import MinkUNet
    model = MinkUNet(num_class=20, cr=0.25)
    seeded_state(model)
    model = model.to(DEV).train()
    coords = t(golden_e2e["coords"])
    batch = {"lidar": SparseTensor(t(golden_e2e["feats"]), coords), "targets": SparseTensor(t(golden_e2e["labels"]), coords)}
    out = model(batch)
    err = np.abs(out["logits"].detach().cpu().numpy() - golden_e2e["logits"]).max()
    assert err < 1e-3, err
    assert abs(float(out["loss"].detach()) - float(golden_e2e["loss"])) < 1e-3
    out["loss"].backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.parameters())


# ---- full-size, size-independent properties (120k-point scan, BASELINE shape) -------------------------
def test_full_scan_properties(hip):
    from openpcseg_amd import functional as F
    from openpcseg_amd.workloads.synthetic import make_batch
    batch = make_batch([0, 1])
    coords = batch["lidar"].C.to(DEV)
    n = coords.shape[0]
    assert n > 150000
    h = hip.hash(coords)
    assert (hip.hash_query(h, h) == torch.arange(n, device=DEV)).all()       # every voxel finds itself
    entry = F.build_kernel_map(coords, coords, (3, 3, 3), (1, 1, 1), (1, 1, 1))
    pairs, koff = entry.fwd.pairs, entry.fwd.koff_host
    assert koff[-1] == pairs.shape[0] and int(entry[1].sum()) == pairs.shape[0]
    centre = pairs[koff[13]:koff[14]]
    assert (centre[:, 0] == centre[:, 1]).all() and centre.shape[0] == n     # convolution_cuda.cu:76-88 assumption
    for k in (0, 5, 26):
        o = pairs[koff[k]:koff[k + 1], 1].long()
        assert (o[1:] > o[:-1]).all()                                        # unique + ascending per offset
    # mirror symmetry of a submanifold map: offset k pairs (i,o) <-> offset 26-k pairs (o,i)
    a = pairs[koff[3]:koff[4]]
    b = pairs[koff[23]:koff[24]]
    assert a.shape == b.shape
    assert (a[torch.argsort(a[:, 0].long())][:, [1, 0]] == b).all()
    # linearity + run-to-run reproducibility of the fused conv
    g = torch.Generator(device=DEV).manual_seed(0)
    x1 = torch.randn(n, 32, device=DEV, generator=g)
    x2 = torch.randn(n, 32, device=DEV, generator=g)
    w = torch.randn(27, 32, 64, device=DEV, generator=g) * 0.05
    y1, y2 = hip.conv_gather_gemm(x1, w, entry.fwd), hip.conv_gather_gemm(x2, w, entry.fwd)
    y12 = hip.conv_gather_gemm(x1 + x2, w, entry.fwd)
    assert ((y1 + y2) - y12).abs().max() <= 1e-4 * y12.abs().max()
    # in-order (ticket) commit: the summation order depends on the map and the tile height only -> bit-reproducible
    assert torch.equal(y1, hip.conv_gather_gemm(x1, w, entry.fwd))
    gw = hip.conv_wgrad(x1, y1, entry.fwd, 0)
    assert torch.equal(gw, hip.conv_wgrad(x1, y1, entry.fwd, 0))
    # <conv(x), g> == <x, dgrad(g)>  (adjointness of fwd / dgrad over the two maps)
    gy = torch.randn(n, 64, device=DEV, generator=g)
    lhs = (y1.double() * gy.double()).sum()
    rhs = (x1.double() * hip.conv_gather_gemm(gy, w.transpose(1, 2).contiguous(), entry.rev).double()).sum()
    assert abs(float(lhs - rhs)) <= 1e-4 * abs(float(lhs))


# ---- fused BatchNorm + residual + ReLU (block fusion above the op boundary) -----------------------------
@pytest.mark.parametrize("c", [32, 96, 384])
def test_bn_relu_gate_bitmask(hip, c):
    """The apply pass packs [y > 0] into n x c/32 words; both backward passes give the same result from the mask as
    from y itself (bit-identical: the gate is the only thing read from either)."""
    torch.manual_seed(c)
    n = 33333
    x = torch.randn(n, c, device=DEV)
    res = torch.randn(n, c, device=DEV)
    w, b = torch.rand(c, device=DEV) + 0.5, torch.randn(c, device=DEV)
    stat = hip.bn_finalize(hip.bn_stats(x), float(n), 1e-5, 0.1, None, None)
    y, mask = hip.bn_apply(x, res, stat, w, b, True, want_mask=True)
    assert torch.equal(y, hip.bn_apply(x, res, stat, w, b, True))
    bits = ((mask.long().unsqueeze(-1) >> torch.arange(32, device=DEV)) & 1).reshape(n, c).bool()
    assert torch.equal(bits, y > 0)
    dy = torch.randn(n, c, device=DEV)
    s_y, s_m = hip.bn_bwd_stats(dy, x, y, stat, True), hip.bn_bwd_stats(dy, x, mask, stat, True)
    assert torch.equal(s_y, s_m)
    for a, bb in zip(hip.bn_bwd_apply(dy, x, y, stat, s_y, float(n), w, True, True),
                     hip.bn_bwd_apply(dy, x, mask, stat, s_y, float(n), w, True, True)):
        assert torch.equal(a, bb)
    with pytest.raises(RuntimeError):  # the mask needs whole 32-channel words
        hip.bn_apply(x[:, :20].contiguous(), None, stat, None, None, True, want_mask=True)


@pytest.mark.parametrize("c,relu,with_res", [(32, True, False), (96, True, True), (256, False, False), (5, True, True),
                                             (384, True, True), (20, True, False)])
def test_fused_batchnorm_matches_torch(hip, c, relu, with_res):
    from openpcseg_amd.fused import FusedBatchNorm
    from openpcseg_amd.sparse import SparseTensor
    torch.manual_seed(c)
    n = 70001
    x = (torch.randn(n, c, device=DEV) * 1.7 + 0.3).requires_grad_(True)
    r = torch.randn(n, c, device=DEV).requires_grad_(True) if with_res else None
    coords = torch.zeros(n, 4, dtype=torch.int32, device=DEV)
    bn = FusedBatchNorm(c).to(DEV).train()
    ref = torch.nn.BatchNorm1d(c).to(DEV).train()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.uniform_(-0.5, 0.5)
        ref.weight.copy_(bn.weight); ref.bias.copy_(bn.bias)
    y = bn(SparseTensor(x, coords), residual=SparseTensor(r, coords) if with_res else None, relu=relu).F
    x2 = x.detach().clone().requires_grad_(True)
    r2 = r.detach().clone().requires_grad_(True) if with_res else None
    t = ref(x2)
    if with_res:
        t = t + r2
    if relu:
        t = torch.relu(t)
    assert (y - t).abs().max() <= 2e-5 * t.abs().max()
    g = torch.randn_like(y)
    y.backward(g)
    t.backward(g)
    assert (x.grad - x2.grad).abs().max() <= 5e-5 * x2.grad.abs().max()
    assert (bn.weight.grad - ref.weight.grad).abs().max() <= 1e-4 * ref.weight.grad.abs().max()
    assert (bn.bias.grad - ref.bias.grad).abs().max() <= 1e-4 * ref.bias.grad.abs().max()
    if with_res:
        assert (r.grad - r2.grad).abs().max() <= 1e-6
    assert torch.allclose(bn.running_mean, ref.running_mean, atol=1e-5) and torch.allclose(bn.running_var, ref.running_var, rtol=1e-4)
    bn.eval(); ref.eval()
    ye = bn(SparseTensor(x.detach(), coords), relu=relu).F
    te = torch.relu(ref(x.detach())) if relu else ref(x.detach())
    assert (ye - te).abs().max() <= 2e-5 * te.abs().max()


@pytest.mark.parametrize("stride", [1, 2, 8])
def test_corner_map_equals_the_op_sequence(hip, stride):
    """pcs_corner_map_f32 == floor / cat / sphash(8 corners) / sphashquery / calc_ti_weights / transposes
    (R:pcseg/model/segmentor/voxel/minkunet/utils.py:69-105): indices bit-exact, weights to the last bit or two."""
    from openpcseg_amd import functional as F
    from openpcseg_amd.sparse import get_kernel_offsets
    rng = np.random.default_rng(stride)
    vox = np.unique(np.concatenate([rng.integers(0, 40, size=(20000, 3)) // stride * stride,
                                    rng.integers(0, 3, size=(20000, 1))], axis=1), axis=0).astype(np.int32)
    pts = np.concatenate([rng.uniform(-1, 41, size=(50000, 3)), rng.integers(0, 3, size=(50000, 1))], axis=1).astype(np.float32)
    xc, zc = t(vox), t(pts)
    idx8, w8 = hip.corner_map(zc, xc, stride)
    corners = get_kernel_offsets(2, stride, 1, device=DEV)
    base = torch.cat([torch.floor(zc[:, :3] / stride).int() * stride, zc[:, -1:].int()], dim=1)
    iq = F.sphashquery(F.sphash(base, corners), F.sphash(xc))
    w = F.calc_ti_weights(zc, iq, scale=stride).t().contiguous()
    assert torch.equal(idx8.long(), iq.t().contiguous())
    assert (w8 - w).abs().max() <= 2e-7
    assert int((idx8 >= 0).sum()) > 1000  # the case really has hits and misses
    assert int((idx8 < 0).sum()) > 1000


def test_fused_linear_matches_torch(hip):
    """The classifier (480 -> 20 over ~1e5..1e6 rows) on the fused conv kernels == nn.Linear, values and gradients."""
    from openpcseg_amd.fused import FusedLinear
    torch.manual_seed(3)
    n = 50021
    lin = FusedLinear(480, 20).to(DEV)
    ref = torch.nn.Linear(480, 20).to(DEV)
    ref.load_state_dict(lin.state_dict())
    x = torch.randn(n, 480, device=DEV, requires_grad=True)
    x2 = x.detach().clone().requires_grad_(True)
    y, t = lin(x), ref(x2)
    assert (y - t).abs().max() <= 2e-5 * t.abs().max()
    g = torch.randn_like(y)
    y.backward(g)
    t.backward(g)
    assert (x.grad - x2.grad).abs().max() <= 2e-5 * x2.grad.abs().max()
    assert (lin.weight.grad - ref.weight.grad).abs().max() <= 2e-5 * ref.weight.grad.abs().max()
    assert (lin.bias.grad - ref.bias.grad).abs().max() <= 2e-5 * ref.bias.grad.abs().max()
    small = torch.randn(100, 480, device=DEV)  # few rows: plain nn.Linear path
    assert torch.allclose(lin(small), ref(small), atol=1e-5)


@pytest.mark.parametrize("amp", [None, torch.bfloat16])
def test_fused_linear_over_column_blocks(hip, amp):
    """forward_parts([z1, z2, z3]) == Linear(cat([z1, z2, z3], 1)): values, the three input gradients, dW, db -- the
    classifier of the workload without the (N, 480) concatenation."""
    from openpcseg_amd.fused import FusedLinear
    torch.manual_seed(4)
    n = 40013
    lin = FusedLinear(480, 20).to(DEV)
    ref = torch.nn.Linear(480, 20).to(DEV)
    ref.load_state_dict(lin.state_dict())
    parts = [torch.randn(n, c, device=DEV, requires_grad=True) for c in (256, 128, 96)]
    parts2 = [p.detach().clone().requires_grad_(True) for p in parts]
    g = torch.randn(n, 20, device=DEV)
    if amp is None:
        y, t, tol = lin.forward_parts(parts), ref(torch.cat(parts2, 1)), 2e-5
    else:
        with torch.autocast("cuda", dtype=amp):
            y = lin.forward_parts(parts)
        t, tol = ref(torch.cat(parts2, 1)), 2e-2
    assert y.shape == t.shape and (y.float() - t).abs().max() <= tol * t.abs().max()
    y.backward(g.to(y.dtype))
    t.backward(g)
    for a, b in zip(parts, parts2):
        assert a.grad.is_contiguous() and (a.grad.float() - b.grad).abs().max() <= tol * b.grad.abs().max()
    assert (lin.weight.grad - ref.weight.grad).abs().max() <= tol * ref.weight.grad.abs().max()
    assert (lin.bias.grad - ref.bias.grad).abs().max() <= tol * ref.bias.grad.abs().max()
    few = [torch.randn(64, c, device=DEV) for c in (256, 128, 96)]  # few rows: the plain path over the concatenation
    assert torch.allclose(lin.forward_parts(few), ref(torch.cat(few, 1)), atol=1e-5)


# ---- device-side sparse_quantize ---------------------------------------------------------
@pytest.mark.parametrize("case", ["scan", "aniso", "ints"])
def test_device_sparse_quantize_matches_reference_golden(hip, case):
    import os
    from openpcseg_amd import hostdata
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "quantize_golden.npz"))
    vs = g[case + "_vs"]
    vs = tuple(float(v) for v in vs) if vs.ndim else float(vs)
    vox, idx, inv = hostdata.sparse_quantize(t(g[case + "_in"]), vs, return_index=True, return_inverse=True)
    assert vox.is_cuda and vox.dtype == torch.int32 and idx.dtype == torch.int64
    assert np.array_equal(vox.cpu().numpy(), g[case + "_vox"])
    assert np.array_equal(idx.cpu().numpy(), g[case + "_idx"]) and np.array_equal(inv.cpu().numpy(), g[case + "_inv"])
    assert torch.equal(hostdata.sparse_quantize(t(g[case + "_in"]), vs), vox)


def test_device_sparse_quantize_full_scan_properties(hip):
    """120k-ray scan at 0.05 m like the dataset transform (semantickitti_voxel.py:112-120): == the oracle, and the
    size-independent properties: keys strictly ascending, index = first row of its voxel, inverse consistent."""
    from openpcseg_amd import hostdata
    from openpcseg_amd.workloads.synthetic import make_scan
    pts = make_scan(seed=1)[:, :3].astype(np.float32)
    pc = np.round(pts / 0.05).astype(np.int32)
    pc -= pc.min(0, keepdims=True)
    vox, idx, inv = hostdata.sparse_quantize(t(pc), return_index=True, return_inverse=True)
    ovox, oidx, oinv = orc.sparse_quantize(pc)
    assert np.array_equal(vox.cpu().numpy(), ovox) and np.array_equal(idx.cpu().numpy(), oidx)
    assert np.array_equal(inv.cpu().numpy(), oinv)
    v = vox.long()
    ext = v.max(0).values + 1
    key = (v[:, 0] * ext[1] + v[:, 1]) * ext[2] + v[:, 2]
    assert bool((key[1:] > key[:-1]).all())
    pcd = t(pc)
    assert torch.equal(pcd[idx], vox) and torch.equal(vox[inv], pcd)
    first = torch.full((vox.shape[0],), pc.shape[0], dtype=torch.int64, device="cuda")
    first.scatter_reduce_(0, inv, torch.arange(pc.shape[0], device="cuda"), reduce="amin")
    assert torch.equal(first, idx)
    # float points + metric voxel size, empty input
    fv = hostdata.sparse_quantize(t(pts), 0.05)
    assert np.array_equal(fv.cpu().numpy(), orc.sparse_quantize(pts, (0.05,) * 3)[0])
    e = hostdata.sparse_quantize(torch.zeros((0, 3), device="cuda"), 0.05, return_index=True)
    assert e[0].shape == (0, 3) and e[1].shape == (0,)


def test_device_sparse_quantize_float64_and_shape(hip):
    """float64 points stay float64 (NumPy divides float64 by float64: a float32 round trip moves points that sit next
    to a voxel face); (n, d != 3) inputs fail like NumPy's broadcast does (TS:torchsparse/utils/quantize.py:33)."""
    from openpcseg_amd import hostdata
    rng = np.random.default_rng(21)
    base = rng.integers(-50, 50, size=(20000, 3)).astype(np.float64) * 0.1
    pts = base + rng.choice([-1e-12, 0.0, 1e-12, 3e-9, -3e-9], size=base.shape)  # within a float32 ulp of the faces
    rvox, ridx, rinv = hostdata.sparse_quantize(pts, 0.1, return_index=True, return_inverse=True)  # NumPy = reference path
    vox, idx, inv = hostdata.sparse_quantize(t(pts), 0.1, return_index=True, return_inverse=True)
    assert np.array_equal(vox.cpu().numpy(), rvox) and np.array_equal(idx.cpu().numpy(), ridx)
    assert np.array_equal(inv.cpu().numpy(), rinv)
    with pytest.raises(ValueError):
        hostdata.sparse_quantize(t(np.zeros((10, 4), np.float32)), 0.1)
    with pytest.raises(RuntimeError):
        hostdata.sparse_quantize(torch.zeros(10, 3), 0.1)  # CPU tensor: no CPU path for tensors


def test_caches_follow_inplace_updates(hip, golden):
    """Hash tables / CSR orders memoised on caller tensors are keyed on (address, shape, in-place version): a static
    buffer refilled with `copy_` or shifted in place must never be served a stale table (the reference recomputes)."""
    from openpcseg_amd import functional as F
    c0 = golden["scene_coords"]
    buf = t(c0).clone()
    e0 = F.build_kernel_map(buf, buf, (3, 3, 3), (1, 1, 1), (1, 1, 1))
    ref0 = orc.build_kmap(c0, c0, 3, 1)
    assert np.array_equal(e0[0].cpu().numpy().astype(np.int64), ref0[0])
    # refill the same storage with a different scene of the same shape
    c1 = c0.copy()
    c1[:, 0] = (c1[:, 0] * 3 + c1[:, 1]) % 97
    c1 = np.unique(c1, axis=0)
    c1 = np.concatenate([c1, c0[: c0.shape[0] - c1.shape[0]] + np.array([[1000, 0, 0, 0]], np.int32)]).astype(np.int32)
    assert c1.shape == c0.shape
    buf.copy_(t(c1))
    e1 = F.build_kernel_map(buf, buf, (3, 3, 3), (1, 1, 1), (1, 1, 1))
    ref1 = orc.build_kmap(c1, c1, 3, 1)
    assert np.array_equal(e1[1].cpu().numpy(), ref1[1]) and np.array_equal(e1[0].cpu().numpy().astype(np.int64), ref1[0])
    # voxelize: the point -> voxel index edited in place between two calls
    rng = np.random.default_rng(2)
    idx = t(rng.integers(0, 50, size=4000).astype(np.int32))
    feats = t(rng.normal(size=(4000, 8)).astype(np.float32))
    for _ in range(2):
        counts = hip.count(idx, 50)
        out = hip.voxelize_fwd(feats, idx, counts)
        close(out, orc.voxelize_fwd(feats.cpu().numpy(), idx.cpu().numpy(), counts.cpu().numpy()), 1e-5)
        idx.copy_(t(rng.integers(0, 50, size=4000).astype(np.int32)))
    # scatter_max: the index tensor refilled
    index = t(rng.integers(0, 30, size=2000).astype(np.int64))
    src = t(rng.normal(size=(2000, 8)).astype(np.float32))
    for _ in range(2):
        out, _ = hip.scatter_max_fwd(src, index, 30)
        ref = orc.scatter_max(src.cpu().numpy(), index.cpu().numpy(), 30)[0]
        assert np.array_equal(out.cpu().numpy(), ref)
        index.copy_(t(rng.integers(0, 30, size=2000).astype(np.int64)))


def test_fused_batchnorm_large_mean_statistics(hip):
    """|mean| >> std: E[x^2] - mean^2 on raw fp32 partial sums would lose var / mean^2 digits. The statistics pass sums
    (x - pivot) and (x - pivot)^2 and un-shifts in double: mean / invstd match a float64 two-pass reference."""
    g = torch.Generator(device=DEV).manual_seed(3)
    n, c = 200000, 64
    x = torch.randn(n, c, device=DEV, generator=g) * 0.01 + torch.linspace(-300.0, 300.0, c, device=DEV)
    sums = hip.bn_stats(x)
    assert float(sums[2 * c]) == n
    stat = hip.bn_finalize(sums, float(n), 1e-5, 0.1, None, None)
    xd = x.double()
    mean, var = xd.mean(0), xd.var(0, unbiased=False)
    assert torch.allclose(stat[:c], mean, rtol=0, atol=1e-6)
    assert torch.allclose(stat[c:], 1.0 / torch.sqrt(var + 1e-5), rtol=2e-4)
    # the device-resident count path (what SyncBN uses) gives the same statistics
    stat2 = hip.bn_finalize(sums, 0.0, 1e-5, 0.1, None, None, count_dev=sums[2 * c:])
    assert torch.equal(stat, stat2)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("c,relu,with_res", [(32, True, False), (96, True, True), (256, False, False), (20, True, True)])
def test_fused_batchnorm_half_io(hip, dtype, c, relu, with_res):
    """FusedBatchNorm over bf16 / fp16 features (mixed precision): statistics and arithmetic in fp32 / double on the
    half-rounded inputs, one rounding at each store -- vs nn.BatchNorm1d in fp32 on the same half-rounded tensors."""
    from openpcseg_amd.fused import FusedBatchNorm
    from openpcseg_amd.sparse import SparseTensor
    g = torch.Generator(device=DEV).manual_seed(c)
    n = 30000
    xh = (torch.randn(n, c, device=DEV, generator=g) * 1.5 + 0.3).to(dtype)
    rh = torch.randn(n, c, device=DEV, generator=g).to(dtype) if with_res else None
    gyh = torch.randn(n, c, device=DEV, generator=g).to(dtype)
    bn = FusedBatchNorm(c).to(DEV).train()
    ref = torch.nn.BatchNorm1d(c).to(DEV).train()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.uniform_(-0.5, 0.5)
        ref.weight.copy_(bn.weight); ref.bias.copy_(bn.bias)
    x1 = xh.clone().requires_grad_(True)
    r1 = rh.clone().requires_grad_(True) if with_res else None
    coords = torch.zeros(n, 4, dtype=torch.int32, device=DEV)
    y = bn(SparseTensor(x1, coords), residual=r1, relu=relu).F
    assert y.dtype == dtype
    y.backward(gyh)
    x2 = xh.float().requires_grad_(True)
    r2 = rh.float().requires_grad_(True) if with_res else None
    yr = ref(x2)
    if with_res:
        yr = yr + r2
    if relu:
        yr = torch.relu(yr)
    yr.backward(gyh.float())
    eps = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    assert (y.float() - yr).abs().max() <= eps * yr.abs().max() + 1e-6
    # the ReLU gate of a value that rounds to zero in half may differ from the fp32 gate: compare where |y| is not tiny
    live = yr.abs() > 4 * eps if relu else torch.ones_like(yr, dtype=torch.bool)
    assert x1.grad.dtype == dtype
    d = (x1.grad.float() - x2.grad).abs()
    assert d[live].max() <= 2 * eps * x2.grad.abs().max() + 1e-6
    if with_res:
        assert (r1.grad.float() - r2.grad).abs()[live].max() <= eps * r2.grad.abs().max() + 1e-6
    assert torch.allclose(bn.weight.grad, ref.weight.grad, rtol=2e-2, atol=2e-2 * float(ref.weight.grad.abs().max()))
    assert torch.allclose(bn.running_var, ref.running_var, rtol=1e-4) and torch.allclose(bn.running_mean, ref.running_mean, atol=1e-5)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_minkunet_step_under_autocast(hip, dtype):
    """One training step of the MinkUNet workload under torch.autocast (what `bench.py --amp` times): the convolutions
    run on the 16-bit MFMA kernels, BatchNorm on half features, master weights / statistics / loss in fp32; the loss
    agrees with the fp32 step to half precision and every parameter receives a finite fp32 gradient."""
    from openpcseg_amd.sparse import SparseTensor
    from openpcseg_amd.workloads.minkunet import MinkUNet
    from openpcseg_amd.workloads.synthetic import make_batch
    from seeded import seeded_state
    b = make_batch([0, 1], n_points=20000)
    coords = b["lidar"].C.to(DEV)

    def batch():
        return {"lidar": SparseTensor(b["lidar"].F.to(DEV), coords), "targets": SparseTensor(b["targets"].F.to(DEV), coords)}
    model = MinkUNet(num_class=20, cr=0.5)
    seeded_state(model)
    model.to(DEV).train()
    l32 = model(batch())["loss"]
    l32.backward()
    g32 = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    model.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=dtype):
        out = model(batch())
    out["loss"].backward()
    assert abs(float(out["loss"]) - float(l32)) <= 0.05 * abs(float(l32)), (float(out["loss"]), float(l32))
    cos = []
    for n, p in model.named_parameters():
        if p.grad is None:
            continue
        assert p.grad.dtype == torch.float32 and torch.isfinite(p.grad).all(), n
        if p.dim() >= 2 and g32[n].abs().max() > 0:
            cos.append(float(torch.nn.functional.cosine_similarity(p.grad.flatten(), g32[n].flatten(), dim=0)))
    assert np.median(cos) > 0.9, np.median(cos)


def test_unique_inverse_csr(hip):
    """torch.unique + sphashquery + spcount of initial_voxelize (utils.py:17-19) from one stable sort: unique hashes
    ascending, point -> voxel map, counts -- bit-exact; and the sort reused as the CSR of spvoxelize."""
    from openpcseg_amd import functional as F
    rng = np.random.default_rng(4)
    cells = np.concatenate([rng.integers(0, 40, size=(200000, 3)), rng.integers(0, 3, size=(200000, 1))], axis=1).astype(np.int32)
    h = hip.hash(t(cells))
    uniq, inv, counts = hip.unique_inverse_csr(h)
    ru, rinv, rc = torch.unique(h, return_inverse=True, return_counts=True)
    assert torch.equal(uniq, ru) and torch.equal(inv, rinv) and torch.equal(counts.long(), rc)
    assert torch.equal(F.sphashquery(h, uniq), inv) and torch.equal(F.spcount(inv.int(), uniq.numel()), counts)
    feats = t(rng.normal(size=(200000, 7)).astype(np.float32))
    out = F.spvoxelize(feats, inv, counts)                      # uses the cached sort
    close(out, orc.voxelize_fwd(feats.cpu().numpy(), inv.cpu().numpy().astype(np.int32), counts.cpu().numpy()), 1e-5)
    e = hip.unique_inverse_csr(torch.zeros(0, dtype=torch.int64, device=DEV))
    assert e[0].numel() == 0 and e[1].numel() == 0 and e[2].numel() == 0


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("c,ct", [(96, 32), (128, 64), (256, 128), (20, 12)])
def test_fused_batchnorm_concat(hip, dtype, c, ct):
    """torchsparse.cat([relu(bn(x)), skip]) of the decoder (R:.../minkunet/minkunet.py:404-416) as ONE apply launch: the
    BN result lands in the left columns of the concat buffer, the skip tensor is copied to the right ones, and backward
    reads its dy through the row stride -- bit-identical to BN followed by torch.cat, forward and backward."""
    from openpcseg_amd.fused import FusedBatchNorm
    from openpcseg_amd.sparse import SparseTensor
    g = torch.Generator(device=DEV).manual_seed(c + ct)
    n = 20000
    x0 = (torch.randn(n, c, device=DEV, generator=g) * 1.5 + 0.3).to(dtype)
    s0 = torch.randn(n, ct, device=DEV, generator=g).to(dtype)
    gy = torch.randn(n, c + ct, device=DEV, generator=g).to(dtype)
    coords = torch.zeros(n, 4, dtype=torch.int32, device=DEV)
    bn = FusedBatchNorm(c).to(DEV).train()
    outs = []
    for fused in (True, False):
        bn.zero_grad(set_to_none=True)
        x = x0.clone().requires_grad_(True)
        s = s0.clone().requires_grad_(True)
        if fused:
            y = bn(SparseTensor(x, coords), relu=True, cat_with=SparseTensor(s, coords)).F
        else:
            y = torch.cat([bn(SparseTensor(x, coords), relu=True).F, s], dim=1)
        assert y.shape == (n, c + ct) and y.dtype == dtype
        (y * gy).sum().backward()
        outs.append((y.detach(), x.grad, s.grad, bn.weight.grad.clone(), bn.bias.grad.clone()))
    for a, b in zip(*outs):
        assert torch.equal(a, b)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_half_paths_empty_and_ragged(hip, dtype):
    """Edge cases of the round-2 kernels: isolated voxels (only the centre offset has pairs), 1 voxel, tile boundary + 1,
    empty tensors -- half conv, wgrad from half operands (wgrad3 and wgrad2<half>), the fp32 split path, conv write-back
    statistics, half BatchNorm, the cylinder front-end and the one-sort voxel set."""
    from openpcseg_amd import cylinder
    from openpcseg_amd import functional as F
    for n in (1, 129, 130):
        c = np.zeros((n, 4), np.int32)
        c[:, 0] = np.arange(n) * 5
        entry = F.build_kernel_map(t(c), t(c), (3, 3, 3), (1, 1, 1), (1, 1, 1))
        rng = np.random.default_rng(n)
        x = torch.from_numpy(rng.normal(size=(n, 128)).astype(np.float32)).to(dtype)
        w = torch.from_numpy((rng.normal(size=(27, 128, 128)) * 0.1).astype(np.float32)).to(dtype)
        xd, wd = x.to(DEV), w.float().to(DEV)
        wp = hip.prepare_weights_h(wd, dtype, transpose=False)
        got = []
        y = hip.conv_gather_gemm_h(xd, wp, 27, 128, entry.fwd, bn_sums=got)
        ref = x.float().numpy() @ w.float().numpy()[13]
        tol = (2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11)
        assert np.abs(y.float().cpu().numpy() - ref).max() <= tol * np.abs(ref).max() + 1e-5
        if got:  # small tiles may not hold the statistics scratch: then the list stays empty
            assert float(got[0][-1]) == n
        gw = hip.conv_wgrad_h(xd, xd, entry.fwd, 0)  # 128 x 128: wgrad3
        close(gw[13], x.float().numpy().T @ x.float().numpy(), 2e-5)
        assert float(gw[0].abs().max()) == 0.0 and float(gw[26].abs().max()) == 0.0
        gw2 = hip.conv_wgrad_h(xd[:, :32].contiguous(), xd[:, :32].contiguous(), entry.fwd, 0)  # 32 x 32: wgrad2<half>
        close(gw2[13], x.float().numpy()[:, :32].T @ x.float().numpy()[:, :32], 2e-5)
        xf = xd.float()
        close(hip.conv_wgrad(xf, xf, entry.fwd, 0, split=True)[13], x.float().numpy().T @ x.float().numpy(), 2e-5)
    # empty tensors
    e4 = torch.zeros(0, 4, dtype=torch.int32, device=DEV)
    entry = F.build_kernel_map(e4, e4, (3, 3, 3), (1, 1, 1), (1, 1, 1))
    wp = hip.prepare_weights_h(torch.zeros(27, 64, 64, device=DEV), dtype, transpose=False)
    assert hip.conv_gather_gemm_h(torch.zeros(0, 64, dtype=dtype, device=DEV), wp, 27, 64, entry.fwd).shape == (0, 64)
    assert float(hip.conv_wgrad_h(torch.zeros(0, 64, dtype=dtype, device=DEV), torch.zeros(0, 64, dtype=dtype, device=DEV),
                                  entry.fwd, 0).abs().max()) == 0.0
    pol, coord, feat = hip.cylinder_partition(torch.zeros(0, 4, device=DEV), [0, -180, -4], [50, 180, 2], [480, 360, 32])
    assert coord.shape == (0, 3) and feat.shape == (0, 9)
    assert cylinder.map_voxel_predictions(torch.randn(5, 20, device=DEV), torch.zeros(0, dtype=torch.int64, device=DEV)).numel() == 0
    with pytest.raises(RuntimeError):
        hip.cylinder_partition(torch.zeros(4, 4, device=DEV), [0, -180, -4], [50, 180, 2], [1, 360, 32])   # grid < 2
    with pytest.raises(RuntimeError):
        hip.prepare_weights_h(torch.zeros(27, 5, 33, device=DEV), dtype, transpose=False)                # shape not served


# ---- criterion tail: Lovasz-softmax in one sort ----------------------------------------------------------------------------
def _lovasz_cases():
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    from make_golden import lovasz_cases
    return lovasz_cases()


def test_lovasz_softmax_golden_and_oracle(hip):
    """pcs_lovasz_softmax_f32 against the reference's own lovasz_softmax (tests/golden/lovasz_golden.npz: value 2e-6,
    gradient 2e-7 with tie groups compared by their total weight) and, element by element INCLUDING the ties, against
    the float64 oracle with a stable sort: the kernel keeps point order inside a tie group like torch's device sort."""
    import os
    from seeded import lovasz_grad_mismatch
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lovasz_golden.npz"))
    for name, p, lab, ign in _lovasz_cases():
        loss, grad = hip.lovasz_softmax(t(p), t(lab), ign)
        lo, go = orc.lovasz_softmax(p, lab, ign)
        assert abs(float(loss) - float(g[name + "_loss"])) <= 2e-6, name
        assert lovasz_grad_mismatch(p, lab, ign, grad.cpu().numpy(), g[name + "_grad"]) <= 2e-7, name
        assert abs(float(loss) - lo) <= 1e-6 and np.abs(grad.cpu().numpy() - go).max() <= 2e-7, name
        value_only, none = hip.lovasz_softmax(t(p), t(lab), ign, need_grad=False)
        assert none is None and float(value_only) == float(loss)
        again, grad2 = hip.lovasz_softmax(t(p), t(lab), ign)
        assert float(again) == float(loss) and torch.equal(grad, grad2)   # no atomics on floats: bit-reproducible


@pytest.mark.parametrize("n,nc,ign", [(300001, 20, 0), (1200000, 20, 0), (70000, 23, 255), (4097, 3, None), (2048, 60, 0)])
def test_lovasz_softmax_large(hip, n, nc, ign):
    """Bench-sized streams (1.2 M points x 19 class slots: several hundred scan blocks per class, a partial last block)
    against the float64 oracle: loss 1e-6, gradient 2e-7 absolute; with saturated rows and an absent class."""
    rng = np.random.default_rng(n)
    z = (rng.normal(size=(n, nc)) * 4).astype(np.float32)
    p = np.exp(z - z.max(1, keepdims=True))
    p = (p / p.sum(1, keepdims=True)).astype(np.float32)
    sat = rng.random(n) < 0.02
    p[sat] = 0.0
    p[sat, rng.integers(0, nc, int(sat.sum()))] = 1.0
    lab = rng.integers(0, nc, n).astype(np.int64)
    lab[lab == nc - 2] = 1
    if ign is not None:
        lab[rng.random(n) < 0.07] = ign
    loss, grad = hip.lovasz_softmax(t(p), t(lab), ign)
    lo, go = orc.lovasz_softmax(p, lab, ign)
    assert abs(float(loss) - lo) <= 1e-6, (float(loss), lo)
    assert np.abs(grad.cpu().numpy() - go).max() <= 2e-7   # (one fp32 rounding of a Jaccard value near 1, over the class count)


def test_lovasz_softmax_edges_and_autograd(hip):
    """No points, only ignored points, a single class; more than 60 classes refused with a message; the autograd
    wrapper of the workload's criterion (`SegLoss`) gives the loss and logits gradient of the torch form."""
    from openpcseg_amd.workloads.losses import SegLoss, lovasz_softmax, lovasz_softmax_device
    loss, grad = hip.lovasz_softmax(torch.zeros((0, 20), device=DEV), torch.zeros(0, dtype=torch.long, device=DEV), 0)
    assert float(loss) == 0.0 and grad.shape == (0, 20)
    p = torch.full((50, 20), 0.05, device=DEV)
    loss, grad = hip.lovasz_softmax(p, torch.zeros(50, dtype=torch.long, device=DEV), 0)
    assert float(loss) == 0.0 and float(grad.abs().max()) == 0.0
    loss, grad = hip.lovasz_softmax(torch.rand(10, 1, device=DEV), torch.zeros(10, dtype=torch.long, device=DEV), 0)
    assert float(loss) == 0.0 and float(grad.abs().max()) == 0.0   # the only class is the ignored one
    with pytest.raises(RuntimeError, match="classes"):
        hip.lovasz_softmax(torch.rand(10, 61, device=DEV), torch.zeros(10, dtype=torch.long, device=DEV), 0)
    torch.manual_seed(3)
    logits = torch.randn(20000, 20, device=DEV)
    target = torch.randint(0, 20, (20000,), device=DEV)
    outs = []
    for fn in (lovasz_softmax_device, lovasz_softmax):
        x = logits.clone().requires_grad_(True)
        val = fn(x.softmax(1), target, ignore=0)
        gx, = torch.autograd.grad(val * 3.0, x)
        outs.append((float(val), gx))
    assert abs(outs[0][0] - outs[1][0]) <= 1e-6 and float((outs[0][1] - outs[1][1]).abs().max()) <= 2e-7
    crit = SegLoss(ignore_index=0, label_smoothing=0.1)
    x = logits.clone().requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        val = crit(x, target)
    val.backward()
    assert torch.isfinite(val) and torch.isfinite(x.grad).all() and x.grad.dtype == torch.float32


def test_classifier_on_the_voxels_equals_the_literal_order(hip, monkeypatch):
    """workloads/minkunet.py forms the class scores on the voxels and devoxelises num_class channels
    (`FusedLinear.devoxelized_part`) instead of devoxelising 480 channels and contracting on the points: the same
    function (interpolation is linear). One training step with dropout ACTIVE (same generator state: the out-of-place
    dropout draws the mask of the in-place one) against PCS_CLASSIFIER_COMMUTE=0: logits 2e-6 of their maximum, loss 1e-6
    relative, every parameter gradient 2e-4 of its abs-sum (fp32 summation order through the whole backward pass)."""
    from openpcseg_amd.sparse import SparseTensor
    from openpcseg_amd.workloads.minkunet import MinkUNet
    from openpcseg_amd.workloads.synthetic import make_batch
    from seeded import seeded_state
    b = make_batch([0, 1], n_points=30000)
    coords = b["lidar"].C.to(DEV)
    model = MinkUNet(num_class=20, cr=0.5)
    seeded_state(model)
    model.to(DEV).train()
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("PCS_CLASSIFIER_COMMUTE", mode)
        model.zero_grad(set_to_none=True)
        torch.manual_seed(7)
        out = model({"lidar": SparseTensor(b["lidar"].F.to(DEV), coords), "targets": SparseTensor(b["targets"].F.to(DEV), coords)})
        out["loss"].backward()
        res[mode] = (out["logits"].detach().float().clone(), float(out["loss"]),
                     {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None})
    (la, lossa, ga), (lb, lossb, gb) = res["0"], res["1"]
    assert float((la - lb).abs().max()) <= 2e-6 * float(la.abs().max())
    assert abs(lossa - lossb) <= 1e-6 * abs(lossa)
    assert ga.keys() == gb.keys()
    for n in ga:
        assert float((ga[n] - gb[n]).abs().sum()) <= 2e-4 * float(ga[n].abs().sum()) + 1e-12, n


def test_bn_reduce_finalize_in_one_launch_and_fp32_sums(hip):
    """`pcs_bn_reduce_partials_finalize` (statistics of a conv write-back -> mean / invstd + running stats, one launch) is
    bit-identical to pcs_bn_reduce_partials + pcs_bn_finalize_f32; the backward reduction leaves [sum g | sum g xhat] in
    fp32 behind the doubles (the parameter gradients without a conversion launch)."""
    rng = np.random.default_rng(5)
    c = random_scene(rng, 60000, 60, 2)
    from openpcseg_amd import functional as F
    from openpcseg_amd.sparse import SparseTensor
    x = SparseTensor(t(rng.normal(size=(c.shape[0], 64)).astype(np.float32) * 3 + 1.5), t(c))
    w = t(rng.normal(size=(27, 64, 96)).astype(np.float32) / 20)
    out = F.conv3d(x, w, 3, bn_stats=True)
    pre = out.bn_sums[0]
    n, ch = out.F.shape
    assert pre.dtype == torch.float64 and pre.numel() % (2 * ch) == 0 and pre.numel() > 2 * ch + 1   # the raw per-tile partials
    rm_a, rv_a = torch.zeros(ch, device=DEV), torch.ones(ch, device=DEV)
    rm_b, rv_b = rm_a.clone(), rv_a.clone()
    stat_a = hip.bn_reduce_finalize(pre, ch, n, 1e-5, 0.1, rm_a, rv_a)
    stat_b = hip.bn_finalize(hip.bn_reduce_partials(pre, ch, n), float(n), 1e-5, 0.1, rm_b, rv_b)
    assert torch.equal(stat_a, stat_b) and torch.equal(rm_a, rm_b) and torch.equal(rv_a, rv_b)
    ref = hip.bn_finalize(hip.bn_stats(out.F), float(n), 1e-5, 0.1, None, None)
    assert float((stat_a - ref).abs().max()) <= 1e-6 * float(ref.abs().max())
    y, gate = hip.bn_apply(out.F, None, stat_a, None, None, True, want_mask=True)
    dy = t(rng.normal(size=(n, ch)).astype(np.float32))
    s2 = hip.bn_bwd_stats(dy, out.F, gate, stat_a, True)
    assert s2.shape == (2 * ch,) and s2._pcs_f32.dtype == torch.float32 and torch.equal(s2._pcs_f32, s2.float())


def test_weights_multi(hip, monkeypatch):
    """csrc/weights_multi.hip: the transposes and half re-packs of a list of layers in one launch are element-wise the
    per-layer entry points' results; `functional._WeightPrep` refreshes every registered copy when a weight's version
    changes and a training step gives bit-identical loss and gradients with the cache on and off (fp32 and bf16)."""
    rng = np.random.default_rng(9)
    shapes = [(27, 96, 96), (8, 128, 96), (27, 56, 112), (1, 256, 128), (27, 32, 64), (5, 40, 72)]
    ws = [t(rng.normal(size=s).astype(np.float32)) for s in shapes]
    jobs, want = [], []
    for w in ws:
        jobs.append((w, hip.prepared_weights_buffer(w, "t", False), "t", False))
        want.append(hip.transpose_weights(w))
        for dt, tr in ((torch.bfloat16, False), (torch.float16, True)):
            k, a, b = w.shape
            if hip.conv_h_applies(*((b, a) if tr else (a, b)), k):
                jobs.append((w, hip.prepared_weights_buffer(w, dt, tr), dt, tr))
                want.append(hip.prepare_weights_h(w, dt, transpose=tr))
    hip.weights_multi(jobs)
    hip.weights_multi(jobs)   # second call: the cached device job table
    assert len(jobs) > len(ws)
    for (w, dst, kind, tr), ref in zip(jobs, want):
        assert dst.shape == ref.shape and torch.equal(dst, ref), (tuple(w.shape), kind, tr)
    from openpcseg_amd.sparse import SparseTensor
    from openpcseg_amd.workloads.minkunet import MinkUNet
    from openpcseg_amd.workloads.synthetic import make_batch
    from seeded import seeded_state
    b = make_batch([0], n_points=20000)
    coords = b["lidar"].C.to(DEV)
    for amp in (None, torch.bfloat16):
        res = {}
        for mode in ("0", "1"):
            monkeypatch.setenv("PCS_WEIGHT_PREP", mode)
            model = MinkUNet(num_class=20, cr=0.5)
            seeded_state(model)
            model.to(DEV).train()
            opt = torch.optim.SGD(model.parameters(), lr=0.05)
            losses = []
            for it in range(3):   # the optimizer steps in between make every cached copy stale
                opt.zero_grad(set_to_none=True)
                torch.manual_seed(it)
                with torch.autocast("cuda", dtype=amp or torch.bfloat16, enabled=amp is not None):
                    out = model({"lidar": SparseTensor(b["lidar"].F.to(DEV), coords), "targets": SparseTensor(b["targets"].F.to(DEV), coords)})
                out["loss"].backward()
                losses.append(float(out["loss"]))
                if it < 2:
                    opt.step()
            res[mode] = (losses, {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None})
        assert res["0"][0] == res["1"][0], (amp, res["0"][0], res["1"][0])
        assert all(torch.equal(res["0"][1][n], res["1"][1][n]) for n in res["0"][1]), amp


@pytest.mark.gpu
@pytest.mark.parametrize("amp", [None, torch.bfloat16], ids=["fp32", "bf16"])
def test_weight_prep_sees_writes_behind_autograd(hip, amp):
    """`w.data.copy_()`, `w.data = t` (EMA / SWA swaps, `module._apply`, raw-pointer optimizers) do not bump
    the parameter's version counter. The prepared copies (`functional._WeightPrep`: dgrad transposes, fragment-ordered half
    weights) belong to one pass over the model and are rebuilt when the next pass starts, so forward AND input gradient follow
    the live weight -- checked against a fresh, cache-less computation after each kind of write."""
    from openpcseg_amd import functional as F
    from openpcseg_amd import modules as spnn
    from openpcseg_amd.sparse import SparseTensor
    from openpcseg_amd.workloads.synthetic import make_batch
    b = make_batch([1], n_points=6000)
    coords = b["lidar"].C.to(DEV)
    torch.manual_seed(0)
    conv = spnn.Conv3d(64, 96, 3).to(DEV)
    assert F._WeightPrep.usable(hip, conv.kernel)
    x0 = torch.randn(coords.shape[0], 64, device=DEV)

    def run(cached):
        os.environ["PCS_WEIGHT_PREP"] = "1" if cached else "0"
        try:
            x = x0.clone().requires_grad_(True)
            with torch.autocast("cuda", dtype=amp or torch.bfloat16, enabled=amp is not None):
                y = conv(SparseTensor(x, coords)).F
            y.float().square().sum().backward()
            return y.detach().float().clone(), x.grad.clone()
        finally:
            os.environ.pop("PCS_WEIGHT_PREP", None)

    def check(what):
        conv.zero_grad(set_to_none=True)
        y1, g1 = run(True)
        y0, g0 = run(False)
        assert torch.equal(y1, y0) and torch.equal(g1, g0), what

    check("first pass")
    check("unchanged weights")
    ver = conv.kernel._version
    conv.kernel.data.mul_(2.0)                       # in place through .data: same version, same storage
    assert conv.kernel._version == ver
    check(".data.mul_")
    conv.kernel.data.copy_(torch.randn_like(conv.kernel) * 0.05)
    check(".data.copy_")
    conv.kernel.data = torch.randn_like(conv.kernel) * 0.05   # new storage, same Parameter object
    check(".data = tensor")
    with torch.no_grad():
        conv.kernel.add_(0.01)                       # the ordinary, version-bumping update
    check("in-place under no_grad")
    F._WEIGHT_PREP.invalidate()
    check("explicit invalidate")


@pytest.mark.gpu
@pytest.mark.parametrize("seeds,npts", [(list(range(12)), None), ([3, 4, 5], 20000), ([7], 5000)])
def test_device_input_pipeline_matches_the_host_dataloader(hip, seeds, npts):
    """SURVEY section 8 f1 with its number (bench.py's `device_input` record): raw (n, 4) scans resident in HBM -> dataset transform
    (R:pcseg/data/dataset/semantickitti/semantickitti_voxel.py:112-120) + sparse_quantize of every frame + sparse_collate_fn in ONE
    device pass (hostdata.sparse_quantize_frames: one sort, one host read per batch) -- bit-exact with the host path
    (NumPy sparse_quantize per frame, TS:torchsparse/utils/quantize.py:24-46, + TS:torchsparse/utils/collate.py:11-59):
    voxel coordinates incl. order and batch column, gathered features, gathered labels."""
    from openpcseg_amd import hostdata
    from openpcseg_amd.workloads import synthetic as syn
    raw = syn.make_raw_batch(seeds, n_points=npts)
    dev_raw = {k: (v.to(DEV) if isinstance(v, torch.Tensor) else v) for k, v in raw.items()}
    d = syn.device_collate(dev_raw)
    b = syn.make_batch(seeds, n_points=npts)
    assert torch.equal(d["lidar"].C.cpu(), b["lidar"].C)
    assert torch.equal(d["lidar"].F.cpu(), b["lidar"].F)
    assert torch.equal(d["targets"].F.cpu(), b["targets"].F)
    # the inverse map sends every point to the voxel that holds its cell
    pc = torch.round(dev_raw["points"][:, :3] / torch.full((), 0.05, device=DEV)).int()
    lo = torch.stack([pc[dev_raw["frames"] == i].min(0).values for i in range(raw["num_frames"])])
    pc = pc - lo[dev_raw["frames"].long()]
    vox, index, inverse = hostdata.sparse_quantize_frames(pc, dev_raw["frames"], raw["num_frames"])
    assert torch.equal(vox[inverse][:, :3], pc) and torch.equal(vox[inverse][:, 3], dev_raw["frames"])
    assert torch.equal(vox, d["lidar"].C)
