"""Host-side logic without a GPU: the C-ABI library loads and exports what the header
declares, the reference import names resolve, containers share their caches, the conv3d
dispatcher / autograd wiring reproduces the reference's outputs when the native backend is
replaced (monkeypatch, test only) by the CPU oracle."""
import os
import re
import sys

import numpy as np
import pytest
import torch

import openpcseg_amd
from openpcseg_amd import functional as F
from openpcseg_amd import hostdata, native
from openpcseg_amd.sparse import PointTensor, SparseTensor, cat, fapply, get_kernel_offsets

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "pcseg_hip.h")).read()
    declared = set(re.findall(r"\b(pcs_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(native.SIGNATURES), declared ^ set(native.SIGNATURES)
    lib = native.load_library()  # dlopen works without a GPU
    for name in declared:
        assert hasattr(lib, name), name
    # and the other way round: nothing named pcs_* leaves the library without a declaration in the header
    import subprocess
    syms = subprocess.run(["nm", "-D", "--defined-only", native.LIB_PATH if hasattr(native, "LIB_PATH") else lib._name],
                          capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (pcs_[a-z0-9_]+)$", syms, re.M))
    assert exported and exported <= declared, exported - declared
    assert lib.pcs_abi_version() == native.ABI_VERSION == 11
    assert lib.pcs_hashtable_capacity(1000) == 2048
    assert lib.pcs_hashtable_bytes(2048) == 2048 * 12
    assert lib.pcs_conv_tile_rows(32, 32) in (64, 128)


def test_no_cpu_fallback():
    be = native.HipBackend()
    with pytest.raises(RuntimeError, match="HIP device tensor"):
        be.hash(torch.zeros(4, 4, dtype=torch.int32))
    with pytest.raises(RuntimeError, match="HIP device tensor"):
        F.spvoxelize(torch.zeros(4, 4), torch.zeros(4, dtype=torch.int32), torch.ones(4, dtype=torch.int32))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "openpcseg_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f
                assert "pcs_oracle" not in src, f
    # the measurement / analysis tools are not checkers either: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
    # worker may touch oracle/ (bench.py: inside _cpu_baseline_worker only)
    for f in sorted(os.listdir(os.path.join(ROOT, "tools"))):
        if f.endswith((".py", ".sh")):
            src = open(os.path.join(ROOT, "tools", f)).read()
            assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M) and "pcs_oracle" not in src, f
    bench = open(os.path.join(ROOT, "bench.py")).read()
    body = bench[bench.index("def _cpu_baseline_worker"):]
    body = body[:body.index("\ndef ", 1)]
    assert len(re.findall(r"^\s*(?:from|import)\s+oracle\b", bench, re.M)) == len(re.findall(r"^\s*(?:from|import)\s+oracle\b", body, re.M)) > 0


def test_product_sources_hold_no_retired_kernels():
    """The column-parallel "ring" kernels of round 4 (0.47-0.70x the wave kernels, profiles/round4_ring.md) are retired: no
    hook, stub header or export of them is left in the product sources, the C ABI or the built library; their text is kept as a
    record under tools/experimental/csrc/*.txt (not compiled by anything)."""
    import subprocess
    csrc = os.path.join(ROOT, "openpcseg_amd", "csrc")
    for f in os.listdir(csrc):
        assert not f.startswith("conv_ring"), f
        assert "conv_ring" not in open(os.path.join(csrc, f)).read() and "PCS_WITH_RING" not in open(os.path.join(csrc, f)).read(), f
    hdr = open(os.path.join(ROOT, "include", "pcseg_hip.h")).read()
    assert "pcs_conv_ring_enable(" not in hdr and "pcs_conv_ring_applies(" not in hdr
    assert all(f.endswith(".txt") for f in os.listdir(os.path.join(ROOT, "tools", "experimental", "csrc")))
    lib = os.path.join(ROOT, "openpcseg_amd", "lib", "libpcseg_hip.so")
    if os.path.exists(lib):
        syms = subprocess.run(["nm", "-D", lib], capture_output=True, text=True).stdout
        assert "pcs_conv_ring" not in syms and "conv_ring6" not in syms


def test_reference_import_names():
    openpcseg_amd.install_as_torchsparse()
    import torchsparse
    import torchsparse.nn as spnn
    import torchsparse.nn.functional as TF
    from torchsparse import PointTensor as PT, SparseTensor as ST  # noqa: F401
    from torchsparse.nn.utils import fapply as fa, get_kernel_offsets as gko  # noqa: F401
    from torchsparse.utils.collate import sparse_collate_fn  # noqa: F401
    from torchsparse.utils.quantize import sparse_quantize  # noqa: F401
    assert torchsparse.__version__ == "1.4.0"
    for n in ["sphash", "sphashquery", "spcount", "spvoxelize", "spdevoxelize", "calc_ti_weights",
              "spdownsample", "conv3d"]:
        assert callable(getattr(TF, n))
    conv = spnn.Conv3d(4, 8, kernel_size=3, stride=1)
    assert conv.kernel.shape == (27, 4, 8) and conv.bias is None
    assert spnn.Conv3d(4, 8, kernel_size=1).kernel.shape == (4, 8)
    assert torchsparse.cat is cat


def test_kernel_offsets_order():
    assert get_kernel_offsets(3)[:4].tolist() == [[-1, -1, -1], [0, -1, -1], [1, -1, -1], [-1, 0, -1]]
    assert get_kernel_offsets(2).tolist() == [[0, 0, 0], [0, 0, 1], [0, 1, 0], [0, 1, 1], [1, 0, 0], [1, 0, 1],
                                              [1, 1, 0], [1, 1, 1]]
    assert get_kernel_offsets((1, 3, 3))[:4].tolist() == [[0, -1, -1], [0, 0, -1], [0, 1, -1], [0, -1, 0]]
    assert (get_kernel_offsets(3, stride=2) == 2 * get_kernel_offsets(3)).all()
    assert get_kernel_offsets(2).dtype == torch.int32


def test_cache_sharing_semantics():
    x = SparseTensor(torch.zeros(3, 2), torch.zeros(3, 4, dtype=torch.int32), 2)
    assert x.s == (2, 2, 2) and x.F is x.feats and x.C is x.coords
    y = fapply(x, torch.relu)
    z = cat([x, y])
    w = x + y
    for t in (y, z, w):
        assert t.cmaps is x.cmaps and t.kmaps is x.kmaps and t.stride == x.stride
    assert z.F.shape == (3, 4)
    p = PointTensor(torch.zeros(3, 2), torch.zeros(3, 4))
    q = p + p
    assert q.idx_query is p.idx_query and q.additional_features is p.additional_features


def test_sparse_quantize_matches_reference(golden):
    for b in range(2):
        vox, idx = hostdata.sparse_quantize(golden["quant_in_%d" % b], return_index=True)
        assert (vox == golden["quant_out_%d" % b]).all() and (idx == golden["quant_idx_%d" % b]).all()
    a = SparseTensor(np.zeros((2, 1), np.float32), np.array([[0, 0, 0], [1, 1, 1]], np.int32))
    b = SparseTensor(np.zeros((1, 1), np.float32), np.array([[2, 2, 2]], np.int32))
    out = hostdata.sparse_collate_fn([{"lidar": a, "n": 1}, {"lidar": b, "n": 2}])
    assert out["lidar"].C.tolist() == [[0, 0, 0, 0], [1, 1, 1, 0], [2, 2, 2, 1]] and out["n"] == [1, 2]


def test_sparse_quantize_host_path_properties():
    """The NumPy path is a stable sort + run flags (like the device path), not np.unique: same outputs as np.unique over the
    ravel hash -- first occurrence, ascending hash order, inverse -- on random clouds with many duplicates, anisotropic voxel
    sizes, negative coordinates, one row and no rows; ravel_hash itself against the mixed-radix definition."""
    rng = np.random.default_rng(7)
    for n, vs in [(0, 1), (1, 0.5), (7, (0.5, 0.25, 1.0)), (5000, 0.37), (5000, (2, 3, 1))]:
        pts = rng.integers(-40, 40, size=(n, 3)).astype(np.float64) * 0.21
        vox, idx, inv = hostdata.sparse_quantize(pts, vs, return_index=True, return_inverse=True)
        cells = np.floor(pts / np.array(vs if isinstance(vs, tuple) else (vs,) * 3)).astype(np.int32)
        if n == 0:
            assert vox.shape == (0, 3) and idx.shape == (0,) and inv.shape == (0,)
            continue
        rel = (cells - cells.min(0)).astype(np.uint64)
        ext = rel.max(0) + np.uint64(1)
        h = (rel[:, 0] * ext[1] + rel[:, 1]) * ext[2] + rel[:, 2]
        assert (hostdata.ravel_hash(cells) == h).all()
        _, i2, v2 = np.unique(h, return_index=True, return_inverse=True)
        assert (idx == i2).all() and (inv == v2).all() and (vox == cells[i2]).all()
        assert (vox[inv] == cells).all()
        only = hostdata.sparse_quantize(pts, vs)
        assert isinstance(only, np.ndarray) and (only == vox).all()
    mixed = hostdata.sparse_collate([SparseTensor(torch.zeros(2, 3), torch.zeros(2, 3)), SparseTensor(torch.ones(1, 3), torch.ones(1, 3))])
    assert mixed.C.dtype == torch.float32 and mixed.C[:, 3].tolist() == [0.0, 0.0, 1.0] and mixed.F.shape == (3, 3)


def test_sparse_quantize_tensor_input_goes_to_the_backend(oracle_backend):
    """torch tensor in -> tensors out through backend().quantize (HIP in the product; the oracle here)."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "quantize_golden.npz"))
    vox, idx, inv = hostdata.sparse_quantize(torch.from_numpy(g["scan_in"]), float(g["scan_vs"]), return_index=True,
                                             return_inverse=True)
    assert torch.equal(vox, torch.from_numpy(g["scan_vox"])) and torch.equal(idx, torch.from_numpy(g["scan_idx"]))
    assert torch.equal(inv, torch.from_numpy(g["scan_inv"]))
    only = hostdata.sparse_quantize(torch.from_numpy(g["ints_in"]), 3)
    assert torch.equal(only, torch.from_numpy(g["ints_vox"]))


# ---- dispatcher + autograd on the oracle backend vs the reference's outputs ---------------------------
@pytest.mark.parametrize("tag,name,ks,stride,transposed", [
    ("conv_k3s1_N", "k3s1", 3, 1, False), ("conv_k2s2_N", "k2s2", 2, 2, False), ("conv_k2s2_T", "k2s2", 2, 2, True)])
def test_conv3d_dispatch_matches_reference(golden, oracle_backend, tag, name, ks, stride, transposed):
    coords = torch.from_numpy(golden["scene_coords"])
    x = torch.from_numpy(golden[tag + "_x"]).requires_grad_(True)
    w = torch.from_numpy(golden[tag + "_w"]).requires_grad_(True)
    if not transposed:
        inp = SparseTensor(x, coords, 1)
        out = F.conv3d(inp, w, ks, stride=stride)
        key = ((1, 1, 1), (ks,) * 3, (stride,) * 3, (1, 1, 1))
        entry = inp.kmaps[key]
        assert (entry[0].long().numpy() == golden["kmap_%s_nbmaps" % name]).all()
        assert (entry[1].numpy() == golden["kmap_%s_nbsizes" % name]).all()
        if stride > 1:
            assert (out.C.numpy() == golden["ds_" + name]).all() and out.s == (2, 2, 2)
    else:
        # a transposed conv reuses the map its down conv built (conv.py:184-192)
        fine = SparseTensor(torch.zeros(coords.shape[0], 8), coords, 1)
        fine.cmaps[(1, 1, 1)] = coords
        down = F.conv3d(fine, torch.zeros(8, 8, 12), ks, stride=stride)
        inp = SparseTensor(x, down.C, down.s)
        inp.cmaps, inp.kmaps = down.cmaps, down.kmaps
        out = F.conv3d(inp, w, ks, stride=stride, transposed=True)
        assert out.s == (1, 1, 1) and out.C is coords
    assert np.allclose(out.F.detach().numpy(), golden[tag + "_y"], rtol=1e-4, atol=1e-5)
    out.F.backward(torch.from_numpy(golden[tag + "_gy"]))
    assert np.allclose(x.grad.numpy(), golden[tag + "_gx"], rtol=1e-4, atol=1e-5)
    assert np.allclose(w.grad.numpy(), golden[tag + "_gw"], rtol=1e-4, atol=1e-4)
    assert out.kmaps is inp.kmaps and out.cmaps is inp.cmaps


@pytest.mark.parametrize("ks", [3, (1, 3, 3), (3, 1, 3)])
def test_submanifold_reverse_map_is_a_mirror(golden, oracle_backend, ks):
    """Submanifold conv + point-symmetric offsets: the input-sorted map comes from the forward map's slices
    (offset k <- offset K-1-k), and equals the map a second probe pass builds."""
    from openpcseg_amd import functional as F
    from openpcseg_amd.sparse import make_ntuple
    c = torch.from_numpy(golden["scene_coords"])
    entry = F.build_kernel_map(c, c, make_ntuple(ks, ndim=3), (1, 1, 1), (1, 1, 1))
    assert entry._mirror
    built = oracle_backend.build_kmap(c, c, -entry._ctx[2])
    assert torch.equal(entry.rev.pairs, built.pairs) and torch.equal(entry.rev.nbsizes, built.nbsizes)
    assert entry.rev.koff_host == built.koff_host
    # a distinct coordinate tensor or an even kernel does not qualify
    assert not F.build_kernel_map(c, c.clone(), (3, 3, 3), (1, 1, 1), (1, 1, 1))._mirror
    assert not F.build_kernel_map(c, c, (2, 2, 2), (1, 1, 1), (1, 1, 1))._mirror


def test_conv3d_channel_mismatch_raises(golden, oracle_backend):
    coords = torch.from_numpy(golden["scene_coords"])
    with pytest.raises(ValueError, match="mismatch"):
        F.conv3d(SparseTensor(torch.zeros(coords.shape[0], 5), coords, 1), torch.zeros(27, 4, 8), 3)


def test_pointwise_conv_bypasses_backend(oracle_backend):
    x = SparseTensor(torch.ones(3, 2), torch.zeros(3, 4, dtype=torch.int32))
    out = F.conv3d(x, torch.ones(2, 5), 1)
    assert out.F.shape == (3, 5) and out.C is x.C and not x.kmaps


def test_minkunet_e2e_matches_reference_logits(golden_e2e, oracle_backend):
    """Our MinkUNet workload (same state_dict layout) on the oracle backend reproduces the
    reference's MinkUNet + reference backend logits: proves the dispatcher, the map reuse by
    transposed convs, initial_voxelize and voxel_to_point follow the reference."""
    from seeded import seeded_state
    from openpcseg_amd.workloads.minkunet import MinkUNet
    torch.manual_seed(0)
    model = MinkUNet(num_class=20, cr=0.25)
    ref_keys = [k for k in golden_e2e["state_keys"].tolist()]
    assert sorted(model.state_dict().keys()) == sorted(ref_keys)
    seeded_state(model)
    model.train()
    batch = {"lidar": SparseTensor(torch.from_numpy(golden_e2e["feats"]), torch.from_numpy(golden_e2e["coords"])),
             "targets": SparseTensor(torch.from_numpy(golden_e2e["labels"]), torch.from_numpy(golden_e2e["coords"]))}
    out = model(batch)
    diff = (out["logits"].detach().numpy() - golden_e2e["logits"])
    assert np.abs(diff).max() < 1e-3, np.abs(diff).max()
    assert abs(float(out["loss"]) - float(golden_e2e["loss"])) < 1e-3
    out["loss"].backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.parameters())


# the reference's own MinkUNet source on this API: tests/test_reference_models.py (CPU oracle and, under -m gpu, HIP)


def test_batched_lovasz_equals_the_per_class_loop():
    """workloads/losses.py: the (C', n)-batched Lovasz-softmax == the reference's per-class loop
    (R:tools/utils/common/lovasz_losses.py:176-204, classes='present'), value and gradient, with an ignored label
    and an absent class."""
    import torch
    from openpcseg_amd.workloads.losses import lovasz_softmax, lovasz_softmax_per_class
    torch.manual_seed(0)
    for n in (1, 7, 5000):
        logits = torch.randn(n, 20, requires_grad=True)
        lab = torch.randint(0, 20, (n,))
        if n > 7:
            lab[lab == 7] = 3
        p = logits.softmax(1)
        a, b = lovasz_softmax(p, lab, ignore=0), lovasz_softmax_per_class(p, lab, ignore=0)
        ga, = torch.autograd.grad(a, logits, retain_graph=True)
        gb, = torch.autograd.grad(b, logits)
        assert abs(float(a.detach()) - float(b.detach())) <= 1e-6 and (ga - gb).abs().max() <= 1e-7
    empty = torch.zeros(4, 20).softmax(1)
    assert float(lovasz_softmax(empty, torch.zeros(4, dtype=torch.long), ignore=0)) == 0.0
    # ignore labels OUTSIDE [0, nc): the ignored points must not make a clamped class "present"
    for ign in (255, -100, -1):
        logits = torch.randn(4000, 20, requires_grad=True)
        lab = torch.randint(0, 18, (4000,))      # classes 18, 19 absent: 19 is where a clamped 255 would land, 0 for negatives
        lab[lab == 0] = 5
        lab[torch.rand(4000) < 0.1] = ign
        p = logits.softmax(1)
        a, b = lovasz_softmax(p, lab, ignore=ign), lovasz_softmax_per_class(p, torch.where(lab == ign, lab, lab), ignore=ign)
        ga, = torch.autograd.grad(a, logits, retain_graph=True)
        gb, = torch.autograd.grad(b, logits)
        assert abs(float(a.detach()) - float(b.detach())) <= 1e-6 and (ga - gb).abs().max() <= 1e-7, (ign, float(a), float(b))


def test_cpu_baseline_worker_times_one_frame():
    """bench.py's cpu_baseline leg (the reference's compiled CPU backend under the workload model, one frame, no
    extrapolation) on a shrunken frame: the record carries value = 1 / (fwd + bwd), cores, kind and the sample text."""
    import json
    import subprocess
    import sys as _sys
    env = dict(os.environ, PCS_CPU_BASELINE_RAYS="1200", OMP_NUM_THREADS="8")
    out = subprocess.run([_sys.executable, os.path.join(ROOT, "bench.py"), "--cpu-baseline-worker", "1"], env=env,
                         capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert rec["kind"] in ("reference", "port") and rec["unit"] == "frames/s" and rec["cores"] in (1, 8)
    assert abs(rec["value"] - 1.0 / rec["seconds_per_frame"]) <= 0.06 * rec["value"] and "1200 rays" in rec["sample"]


def test_launch_shape_helpers_are_host_functions():
    """The per-layer launch-shape entry points run on the host (no GPU needed): tile heights of the bench shapes
    (profiles/round2_tile_sweep_*.md), which kernels read a tile order, which emit BatchNorm partials."""
    lib = native.load_library()
    pick = lib.pcs_conv_pick_tile_rows_dt
    assert pick(1158864, 5112372, 27, 96, 96, 0) == 384 and pick(1158864, 5112372, 27, 96, 96, 1) == 192
    assert pick(329421, 2752033, 27, 128, 128, 0) == 288 and pick(329421, 2752033, 27, 128, 128, 1) == 144
    # 256 -> 256 under autocast: the small stride-16 level takes the weight-stationary kernel's 4-wave tile [r6], the big stride-8
    # level stays on the tall 8-wave tile of conv_os5h
    assert pick(36068, 331722, 27, 256, 256, 0) == 288 and pick(36068, 331722, 27, 256, 256, 2) == 144
    assert pick(113008, 1001308, 27, 256, 256, 1) in (224, 288) and pick(113008, 1001308, 27, 384, 256, 1) == 144
    assert pick(36068, 331722, 27, 16, 32, 0) == 128 and pick(0, 0, 27, 96, 96, 0) == 128
    assert lib.pcs_conv_pick_tile_rows(113008, 1001308, 27, 256, 256) == pick(113008, 1001308, 27, 256, 256, 0)
    for t in (pick(n, p, 27, ci, co, d) for n, p in ((1158864, 5112372), (113008, 1001308), (3000, 9000))
              for ci, co in ((32, 32), (64, 64), (96, 96), (128, 96), (256, 256), (384, 256)) for d in (0, 1)):
        assert 16 <= t <= 512 and t % 16 == 0
    assert lib.pcs_conv_uses_tile_order(96, 96, 27, 0) == 1 and lib.pcs_conv_uses_tile_order(32, 32, 27, 0) == 1
    assert lib.pcs_conv_uses_tile_order(16, 32, 27, 0) == 0 and lib.pcs_conv_uses_tile_order(4, 32, 27, 1) == 0
    assert lib.pcs_conv_emits_bn_partials(128, 128, 27, 288, 0) == 1 and lib.pcs_conv_emits_bn_partials(96, 96, 27, 384, 0) == 1
    assert lib.pcs_conv_emits_bn_partials(5, 33, 27, 128, 0) == 0
    # BASELINE config 5 (RPVNet cr 1.75) widths are served by the wave kernels in fp32 AND in half (round 3: TAIL instances)
    for ci, co in ((56, 56), (56, 112), (112, 112), (224, 168), (168, 168), (336, 224), (224, 448), (672, 448), (448, 448)):
        assert lib.pcs_conv_uses_tile_order(ci, co, 27, 0) == 1 and lib.pcs_conv_uses_tile_order(ci, co, 27, 1) == 1, (ci, co)
        assert lib.pcs_conv_h_applies(ci, co, 27) == 1 and lib.pcs_conv_h_applies(co, ci, 8) == 1
        assert lib.pcs_conv_prepared_weights_bytes(27, ci, co) > 0
    assert lib.pcs_conv_h_applies(100, 64, 27) == 0 and lib.pcs_conv_uses_tile_order(100, 64, 27, 0) == 1  # cin % 4: fp32 only
    assert lib.pcs_conv_h_applies(51, 51, 27) == 0 and lib.pcs_conv_uses_tile_order(51, 51, 27, 0) == 0   # cr 1.6: generic kernel



def test_comm_overlap_summary_on_synthetic_trace():
    """bench.py's `comm` record: RCCL kernel time per step, the part that ran beside compute kernels, and whether the first
    all-reduce kernel started before the last backward convolution finished -- on a hand-made device trace."""
    import types
    sys.path.insert(0, ROOT)
    import bench

    def ev(name, t0, t1):
        return types.SimpleNamespace(name=name, device_type="DeviceType.CUDA", time_range=types.SimpleNamespace(start=t0, end=t1))
    events = [ev("conv_os5_kernel<4>", 0, 1000), ev("wgrad2_kernel", 1000, 1800), ev("ncclDevKernel_AllReduce_Sum_f32", 900, 1500),
              ev("conv_os5_kernel<6>", 1800, 2600), ev("ncclDevKernel_AllReduce_Sum_f32", 2500, 3100), ev("multi_tensor_apply", 3100, 3200),
              types.SimpleNamespace(name="cpu_op", device_type="DeviceType.CPU", time_range=types.SimpleNamespace(start=0, end=5000))]
    rec = bench.comm_overlap_summary(events)
    assert rec["rccl_kernels"] == 2 and abs(rec["rccl_ms_per_step"] - 1.2) < 1e-9
    assert abs(rec["overlapped_with_compute_ms"] - 0.7) < 1e-9 and abs(rec["exposed_ms"] - 0.5) < 1e-9   # 600 + 100 us beside compute
    assert rec["first_rccl_kernel_before_last_conv_ends"] is True and abs(rec["first_rccl_to_last_conv_end_ms"] - 1.7) < 1e-9
    assert bench.comm_overlap_summary([events[0]])["rccl_kernels"] == 0


def test_bench_line_fits_the_drivers_tail():
    """bench.py's compact line must stay under the 2 000 characters the driver keeps (it keeps the TAIL: a longer line loses metric /
    value). The committed line fits untouched; an over-long one is trimmed text first, secondary records next, headline keys never."""
    import copy
    import json
    sys.path.insert(0, ROOT)
    import bench
    log = os.path.join(ROOT, "profiles", "round5_bench.log")
    if not os.path.exists(log):
        pytest.skip("no committed bench line")
    raw = [l for l in open(log).read().splitlines() if l.startswith("{")][-1]
    res = json.loads(raw)
    keep = copy.deepcopy(res)
    line = bench.fit_line(res)
    assert json.loads(line) == keep and "trimmed" not in res and len(line) <= bench.LINE_LIMIT    # untouched
    big = copy.deepcopy(keep)
    big["cpu_baseline"]["sample"] = "worker failed: " + "x" * 400
    big["models"].update({"model%d" % i: [1.0, 2.0, 3.0, 4.0] for i in range(20)})
    line = bench.fit_line(big)
    out = json.loads(line)
    assert len(line) <= bench.LINE_LIMIT and out["trimmed"] is True
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "dtype", "roofline"):
        assert out[k] == keep[k]
    assert out["cpu_baseline"]["value"] == keep["cpu_baseline"]["value"] and out["cpu_baseline"]["kind"] == "reference"
    assert out["amp_bf16"]["value"] == keep["amp_bf16"]["value"]          # numbers survive; labels and the long lists went first


def test_arithmetic_policies_are_opt_in():
    """The fp32 library path computes in fp32 MFMA arithmetic unless a caller selects a split policy explicitly."""
    from openpcseg_amd import functional as F
    assert F.get_conv_policy() == "fp32" and F.get_wgrad_policy() == "fp32"
    for setter in (F.set_conv_policy, F.set_wgrad_policy):
        with pytest.raises(ValueError):
            setter("tf32")
    F.set_conv_policy("bf16x3")
    F.set_wgrad_policy("bf16x3")
    try:
        assert F.get_conv_policy() == "bf16x3" and F._wgrad_split(96, 96) and not F._wgrad_split(64, 96)
    finally:
        F.set_conv_policy("fp32")
        F.set_wgrad_policy("fp32")
    lib = native.load_library()
    assert lib.pcs_conv_x3_applies(96, 96, 27) == 1 and lib.pcs_conv_x3_applies(56, 448, 27) == 1
    assert lib.pcs_conv_x3_applies(4, 32, 27) == 0 and lib.pcs_conv_x3_applies(96, 16, 27) == 0 and lib.pcs_conv_x3_applies(100, 64, 27) == 0
    assert lib.pcs_conv_x3_column_tiles(96) == 6 and lib.pcs_conv_x3_column_tiles(256) == 4 and lib.pcs_conv_x3_column_tiles(32) == 2
    assert lib.pcs_conv_prepared_weights_x3_bytes(27, 96, 96) == 3 * 27 * 6 * 3 * 1024


def test_weights_multi_plan_is_a_host_function():
    """pcs_weights_multi_plan fills the launch geometry of a job table without a device: work-block prefix sums for transposes
    (one 32 x 32 tile per block) and half re-packs (256 fragment words per block), and refuses shapes the half kernels do not serve."""
    import ctypes
    from openpcseg_amd import native
    lib = native.load_library()
    arr = (native._WeightJob * 3)()
    for j, (k, a, b, kind, tr) in zip(arr, ((27, 96, 96, 0, 0), (8, 128, 96, 1, 1), (27, 56, 112, 2, 0))):
        j.src, j.dst, j.K, j.A, j.B, j.kind, j.transpose = 4096, 8192, k, a, b, kind, tr
    blocks = lib.pcs_weights_multi_plan(ctypes.byref(arr), 3)
    t0 = 27 * 3 * 3
    assert arr[0].first_block == 0 and arr[1].first_block == t0
    # job 1: dgrad of 128 -> 96: contraction over B = 96 (3 steps), columns A = 128
    assert arr[1].ns == 3 and arr[1].nt16 * 16 >= 128 and arr[1].nctt in (2, 4, 6, 8)
    w1 = (8 * arr[1].nt16 * arr[1].ns * 64 + 255) // 256
    assert arr[2].first_block == t0 + w1 and arr[2].ns == 2   # 56 channels: two 32-channel steps, the second zero-padded
    assert blocks == arr[2].first_block + (27 * arr[2].nt16 * arr[2].ns * 64 + 255) // 256
    arr[1].A = 30   # contraction / columns the half kernels do not serve
    assert lib.pcs_weights_multi_plan(ctypes.byref(arr), 3) == -1 and b"job 1" in lib.pcs_last_error()


def test_weight_prep_cache_refreshes_all_layers_once(monkeypatch):
    """functional._WeightPrep: the prepared copies belong to one PASS over the model. The first request of a new pass (= a copy
    that was already handed out is asked for again) refreshes every registered layer's copies in ONE backend call, whatever
    happened to the weights in between -- including writes the version counter does not see (`w.data.copy_()`); within a
    pass a version bump or a new storage also refreshes; a dead parameter's copies are dropped."""
    import torch
    from openpcseg_amd import functional as F

    class FakeBackend:
        def __init__(self):
            self.calls = []

        def prepared_weights_buffer(self, weight, kind, transpose):
            return torch.zeros(1)

        def weights_multi(self, jobs):
            self.calls.append([(w.data_ptr(), kind, bool(tr)) for w, dst, kind, tr in jobs])

    monkeypatch.setattr(F._WeightPrep, "usable", staticmethod(lambda be, w: True))
    prep, be = F._WeightPrep(), FakeBackend()
    ws = [torch.nn.Parameter(torch.randn(27, 8, 8)) for _ in range(3)]
    for w in ws:                      # first pass: every layer is new -> one call each (nothing else is stale yet)
        prep.get(be, w, ("t",))
    assert [len(c) for c in be.calls] == [1, 1, 1]
    be.calls.clear()
    for w in ws:                      # second pass, weights untouched: ONE call for all three, at the first request
        prep.get(be, w, ("t",))
    assert [len(c) for c in be.calls] == [3]
    be.calls.clear()
    ws[1].data.mul_(2.0)              # behind autograd's back: same version counter, same storage
    prep.get(be, ws[2], ("t",))       # third pass starts at another layer ...
    assert len(be.calls) == 1 and len(be.calls[0]) == 3   # ... and refreshes all three, the silently rewritten one included
    prep.get(be, ws[1], ("t",))
    assert len(be.calls) == 1         # same pass: a hit
    with torch.no_grad():
        ws[0].add_(1.0)               # a versioned write in the middle of a pass (ws[0] not handed out yet in this pass)
    prep.get(be, ws[0], ("t",))
    assert len(be.calls) == 2 and be.calls[1] == [(ws[0].data_ptr(), "t", False)]
    prep.get(be, ws[1], (torch.bfloat16, True))           # a new kind for one layer: only that copy
    assert len(be.calls) == 3 and be.calls[2] == [(ws[1].data_ptr(), torch.bfloat16, True)]
    ws[2].data = torch.randn(27, 8, 8)                    # new storage for a parameter of this pass ...
    prep.invalidate()                                     # ... and the explicit switch: everything is stale
    be.calls.clear()
    prep.get(be, ws[0], ("t",))
    assert len(be.calls) == 1 and len(be.calls[0]) == 4   # three transposes + the bf16 copy
    assert (ws[2].data_ptr(), "t", False) in be.calls[0]
    del ws[0]
    import gc
    gc.collect()
    fresh = torch.nn.Parameter(torch.randn(8, 4, 4))
    prep.get(be, fresh, ("t",))       # registering a new parameter drops the dead one's copies
    assert sum(1 for v in prep.entries.values() if v[0]() is not None) == len(prep.entries) == 3


def test_step_budget_tools(tmp_path):
    """tools/stats_diff.py + tools/step_budget.py: the difference of two kernel-statistics files grouped by owner."""
    import subprocess
    import sys as _sys
    a, b, d = tmp_path / "a.csv", tmp_path / "b.csv", tmp_path / "d.csv"
    head = '"Name","Calls","TotalDurationNs","AverageNs"\n'
    a.write_text(head + '"void (anonymous namespace)::conv_os5_kernel<4>(pcs::ConvArgs)",30,30000000,1000000\n"__amd_rocclr_copyBuffer",600,2400000,4000\n')
    b.write_text(head + '"void (anonymous namespace)::conv_os5_kernel<4>(pcs::ConvArgs)",70,70000000,1000000\n"__amd_rocclr_copyBuffer",680,2720000,4000\n'
                        '"void (anonymous namespace)::bn_apply_kernel<4, F32>(void const*)",252,12600000,50000\n')
    tools = os.path.join(ROOT, "tools")
    subprocess.run([_sys.executable, os.path.join(tools, "stats_diff.py"), str(a), str(b), str(d)], check=True)
    out = subprocess.run([_sys.executable, os.path.join(tools, "step_budget.py"), str(d), "4", "t"], check=True, capture_output=True, text=True).stdout
    assert "| conv fwd / dgrad (fused gather-GEMM-scatter) | 10 | 10.00 |" in out
    assert "| copies / fills (runtime) | 20 | 0.08 |" in out and "| BatchNorm" in out and "| **total** | **93** |" in out


def test_sparse_quantize_frames_host_logic(monkeypatch):
    """hostdata.sparse_quantize_frames / workloads.synthetic.device_collate: the key construction -- one
    lexicographic key with the frame on top orders every frame like its own ravel hash -- against the per-frame NumPy path
    (make_batch), with the backend's sort / flag / scan / emit pass restated in torch (the HIP pass itself: -m gpu,
    test_device_input_pipeline_matches_the_host_dataloader). Equal-length, ragged and single-frame batches."""
    import torch
    from openpcseg_amd import native
    from openpcseg_amd.workloads import synthetic as syn

    class TorchPass:
        def quantize_sorted_keys(self, keys, coords, frames):
            skeys, perm = torch.sort(keys, stable=True)
            flags = torch.ones_like(skeys)
            flags[1:] = (skeys[1:] != skeys[:-1]).long()
            rank = torch.cumsum(flags, 0)
            index = perm[flags.bool()]
            inverse = torch.empty_like(keys)
            inverse[perm] = rank - 1
            return torch.cat([coords[index], frames[index].int()[:, None]], 1), index, inverse

    monkeypatch.setattr(native, "_BACKEND", TorchPass())
    monkeypatch.setattr(torch.Tensor, "is_cuda", property(lambda self: True))   # the entry point refuses host tensors: no CPU path
    for seeds, npts in (([3, 4, 5], 6000), ([7], 3000)):
        d = syn.device_collate(syn.make_raw_batch(seeds, n_points=npts))
        b = syn.make_batch(seeds, n_points=npts)
        assert torch.equal(d["lidar"].C, b["lidar"].C) and torch.equal(d["lidar"].F, b["lidar"].F)
        assert torch.equal(d["targets"].F, b["targets"].F)
    raw = syn.make_raw_batch([3, 4], n_points=4000)           # ragged: the second frame keeps 2 500 of its rows
    keep = torch.cat([torch.arange(0, 4000), torch.arange(4000, 6500)])
    rag = {"points": raw["points"][keep], "frames": raw["frames"][keep], "labels": raw["labels"][keep], "num_frames": 2,
           "offsets": [0, 4000, 6500]}
    d = syn.device_collate(rag)
    from openpcseg_amd.hostdata import sparse_collate_fn
    frames = [syn.voxelize_scan(rag["points"][a:b].numpy(), seed=0) for a, b in ((0, 4000), (4000, 6500))]
    ref = sparse_collate_fn(frames)
    assert torch.equal(d["lidar"].C, ref["lidar"].C) and torch.equal(d["lidar"].F, ref["lidar"].F)
