#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE in this container.

Needs /root/reference (so it cannot run on the GPU box; the .npz files are committed).
  - the reference's Python package (package/torchsparse.zip, unzipped to a temp dir) is
    imported on top of its own compiled CPU backend (oracle/_ref, built by
    oracle/build_ref.py from the reference's C++ sources);
  - for the end-to-end vector, the reference's MinkUNet (pcseg/model/segmentor/voxel/minkunet)
    is imported unmodified with import stubs for packages that are absent here.
Known reference CPU-twin defects are worked around WITHOUT changing semantics:
  - kernel_hash_cpu uses row 0's batch index for every row (hash_cpu.cpp:29): called per batch;
  - devoxelize_backward_cpu is wrong (devoxelize_cpu.cpp:51-53): replaced by the restatement of
    devoxelize_cuda.cu:37-57 where a golden needs a backward pass (main_full).
Usage: python tests/golden/make_golden.py [models | quantize | lovasz | config2 | config3 | config4 | config5 | cylinder]
"""
import os
import sys
import tempfile
import types
import zipfile
import zlib

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(OUT))
sys.path.insert(0, OUT)
from seeded import seeded_state  # noqa: E402


def import_reference_torchsparse():
    from oracle import build_ref
    build_ref.build()
    backend = build_ref.load()
    tmp = tempfile.mkdtemp(prefix="pcs_golden_")
    with zipfile.ZipFile("/root/reference/package/torchsparse.zip") as zf:
        zf.extractall(tmp, [m for m in zf.namelist()
                            if m.startswith("torchsparse/torchsparse/") and m.endswith(".py")])
    sys.path.insert(0, os.path.join(tmp, "torchsparse"))
    sys.modules["torchsparse.backend"] = backend
    import torchsparse  # the REFERENCE package
    torchsparse.backend = backend
    assert torchsparse.__version__ == "1.4.0"
    return torchsparse


def main():
    ts = import_reference_torchsparse()
    import torchsparse.nn.functional as F
    from torchsparse.nn.utils import get_kernel_offsets
    from torchsparse.utils.quantize import sparse_quantize
    from openpcseg_amd.workloads.synthetic import make_scan

    g = {}
    rng = np.random.default_rng(7)

    # ---- K1/K2 hashes (incl. negative coords, several batches) -----------------------------------
    kat = torch.tensor([[0, 0, 0, 0], [1, 2, 3, 0], [-1, 5, 7, 1], [100, 200, 30, 3]], dtype=torch.int32)
    g["kat_coords"], g["kat_hash"] = kat.numpy(), F.sphash(kat).numpy()
    c = torch.from_numpy(np.concatenate([rng.integers(-300, 300, size=(500, 3)),
                                         rng.integers(0, 3, size=(500, 1))], axis=1).astype(np.int32))
    c[0, :3] = torch.tensor([2 ** 31 - 1, -2 ** 31, -1])
    g["hash_coords"], g["hash_out"] = c.numpy(), F.sphash(c).numpy()
    off = get_kernel_offsets(3, 2)
    kh = torch.zeros(off.shape[0], c.shape[0], dtype=torch.long)
    for b in range(3):  # per batch: works around hash_cpu.cpp:29
        sel = (c[:, 3] == b).nonzero().squeeze(1)
        kh[:, sel] = F.sphash(c[sel].contiguous(), off)
    g["khash_offsets"], g["khash_out"] = off.numpy(), kh.numpy()

    # ---- a small voxelised scene (2 frames) ---------------------------------------------------------
    frames = []
    for b in range(2):
        pts = make_scan(seed=b, n_points=1500)
        pc = np.round(pts[:, :3] / 0.2).astype(np.int32)
        pc -= pc.min(0, keepdims=1)
        vox, inds = sparse_quantize(pc, return_index=True)
        g["quant_in_%d" % b], g["quant_out_%d" % b], g["quant_idx_%d" % b] = pc, vox, inds
        frames.append(np.concatenate([vox, np.full((vox.shape[0], 1), b, np.int32)], axis=1))
    # hash order (what initial_voxelize produces) for stride 1
    coords = torch.from_numpy(np.concatenate(frames).astype(np.int32))
    h = F.sphash(coords)
    coords = coords[torch.argsort(h)].contiguous()
    g["scene_coords"] = coords.numpy()

    # ---- hash query --------------------------------------------------------------------------------
    ref_h = F.sphash(coords)
    q = torch.cat([ref_h[::3], ref_h[:50] + 1])
    g["query_q"], g["query_out"] = q.numpy(), F.sphashquery(q, ref_h).numpy()

    # ---- spdownsample (fast + general branches) and kernel maps -------------------------------------
    def per_batch_kmap(inc, outc, ks, in_stride):
        """reference conv.py:156-176, with kernel_hash called per batch (CPU twin bug)."""
        offsets = get_kernel_offsets(ks, stride=in_stride)
        references = F.sphash(inc)
        queries = torch.zeros(offsets.shape[0], outc.shape[0], dtype=torch.long)
        for b in outc[:, 3].unique().tolist():
            sel = (outc[:, 3] == b).nonzero().squeeze(1)
            queries[:, sel] = F.sphash(outc[sel].contiguous(), offsets)
        results = F.sphashquery(queries, references)
        nbsizes = torch.sum(results != -1, dim=1)
        nbmaps = torch.nonzero(results != -1)
        nbmaps[:, 0] = results.view(-1)[nbmaps[:, 0] * results.size(1) + nbmaps[:, 1]]
        return nbmaps, nbsizes

    cases = [("k3s1", 3, 1, 1), ("k2s2", 2, 2, 1), ("k133", (1, 3, 3), 1, 1), ("k313", (3, 1, 3), 1, 1),
             ("k3s2", 3, 2, 1), ("k3s221", 3, (2, 2, 1), 1)]
    cur = {1: coords}
    for name, ks, st, ts_ in cases:
        inc = cur[1]
        st3 = (st,) * 3 if isinstance(st, int) else st
        if all(s == 1 for s in st3):
            outc = inc
        else:
            outc = F.spdownsample(inc, st, ks, ts_)
            g["ds_%s" % name] = outc.numpy()
        nbmaps, nbsizes = per_batch_kmap(inc, outc, ks, ts_)
        g["kmap_%s_nbmaps" % name], g["kmap_%s_nbsizes" % name] = nbmaps.numpy(), nbsizes.numpy()
    # second level: tensor stride 2 -> 4 (k2 s2) and k3 at stride 2
    c2 = torch.from_numpy(g["ds_k2s2"])
    c4 = F.spdownsample(c2, 2, 2, 2)
    g["ds_k2s2_l2"] = c4.numpy()
    nbmaps, nbsizes = per_batch_kmap(c2, c2, 3, 2)
    g["kmap_k3s1_l2_nbmaps"], g["kmap_k3s1_l2_nbsizes"] = nbmaps.numpy(), nbsizes.numpy()

    # ---- convolution fwd / bwd through the reference autograd Function ---------------------------
    from torchsparse.nn.functional.conv import ConvolutionFunction
    tg = torch.Generator().manual_seed(3)
    for name, cin, cout, transposed in [("k3s1", 16, 24, False), ("k2s2", 8, 12, False), ("k2s2", 12, 8, True)]:
        nbmaps = torch.from_numpy(g["kmap_%s_nbmaps" % name])
        nbsizes = torch.from_numpy(g["kmap_%s_nbsizes" % name])
        n_in = coords.shape[0]
        n_out = g["ds_%s" % name].shape[0] if ("ds_%s" % name) in g else n_in
        k = nbsizes.shape[0]
        x = torch.randn(n_out if transposed else n_in, cin, generator=tg, requires_grad=True)
        w = (torch.randn(k, cin, cout, generator=tg) * 0.2).requires_grad_(True)
        y = ConvolutionFunction.apply(x, w, nbmaps, nbsizes, (n_in, n_out), transposed)
        gy = torch.randn(y.shape, generator=tg)
        y.backward(gy)
        tag = "conv_%s_%s" % (name, "T" if transposed else "N")
        for key, val in [("x", x), ("w", w), ("y", y), ("gy", gy), ("gx", x.grad), ("gw", w.grad)]:
            g["%s_%s" % (tag, key)] = val.detach().numpy()

    # ---- voxelize / devoxelize / trilinear weights ---------------------------------------------------
    npts = 800
    idx = torch.from_numpy(rng.integers(0, 300, size=npts).astype(np.int32))
    counts = F.spcount(idx, 300)
    feats = torch.randn(npts, 6, generator=tg)
    g["vox_idx"], g["vox_counts"], g["vox_feats"] = idx.numpy(), counts.numpy(), feats.numpy()
    g["vox_out"] = ts.backend.voxelize_forward_cpu(feats, idx, counts).numpy()
    g["vox_bwd"] = ts.backend.voxelize_backward_cpu(torch.from_numpy(g["vox_out"]), idx, counts, npts).numpy()
    pcoords = torch.cat([torch.rand(npts, 3, generator=tg) * 40, torch.zeros(npts, 1)], dim=1)
    idxq = torch.from_numpy(rng.integers(-1, 300, size=(8, npts)).astype(np.int64))
    for scale in (1, 2, 4):
        g["tiw_s%d" % scale] = F.calc_ti_weights(pcoords, idxq, scale=scale).numpy()
    g["tiw_coords"], g["tiw_idxq"] = pcoords.numpy(), idxq.numpy()
    w8 = torch.from_numpy(g["tiw_s2"]).t().contiguous()
    vf = torch.randn(300, 10, generator=tg)
    g["devox_feat"] = vf.numpy()
    g["devox_out"] = ts.backend.devoxelize_forward_cpu(vf, idxq.t().contiguous().int(), w8).numpy()

    np.savez_compressed(os.path.join(OUT, "ops_golden.npz"), **g)
    print("wrote ops_golden.npz with", len(g), "arrays")

    # ---- end to end: the reference's own MinkUNet on the reference backend ----------------------------
    e2e = make_e2e(ts)
    np.savez_compressed(os.path.join(OUT, "minkunet_e2e_golden.npz"), **e2e)
    print("wrote minkunet_e2e_golden.npz; logits", e2e["logits"].shape)


class _AttrDict(dict):
    def __getattr__(self, k):
        try:
            v = self[k]
        except KeyError:
            raise AttributeError(k)
        return _AttrDict(v) if isinstance(v, dict) and not isinstance(v, _AttrDict) else v

    __setattr__ = dict.__setitem__


def import_reference_minkunet():
    for name in ["torch_scatter", "range_utils", "range_utils.nn", "range_utils.nn.functional",
                 "rangelib_cuda", "SharedArray", "cv2", "torchvision", "torchvision.transforms",
                 "torchvision.transforms.functional", "prettytable", "matplotlib", "matplotlib.pyplot",
                 "torchvision.models", "torchvision.models.resnet", "numba"]:
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    ed = types.ModuleType("easydict")
    ed.EasyDict = _AttrDict
    sys.modules.setdefault("easydict", ed)
    from stage_reference import reference_root
    root = reference_root()
    if root is None:
        raise RuntimeError("neither /root/reference nor the staged tests/_refsrc is present")
    if root not in sys.path:
        sys.path.insert(0, root)
    import importlib
    return importlib.import_module("pcseg.model.segmentor.voxel.minkunet.minkunet")


def make_e2e(ts):
    from openpcseg_amd.workloads.synthetic import make_batch
    mod = import_reference_minkunet()
    cfg = _AttrDict(NAME="MinkUNet", IGNORE_LABEL=0, IN_FEATURE_DIM=4, BLOCK="ResBlock",
                    NUM_LAYER=[2, 3, 4, 6, 2, 2, 2, 2], PLANES=[32, 32, 64, 128, 256, 256, 128, 96, 96],
                    cr=0.25, DROPOUT_P=0.0, LABEL_SMOOTHING=0.1, IF_DIST=False)
    torch.manual_seed(0)
    model = mod.MinkUNet(cfg, 20)
    seeded_state(model)
    model.train()  # batch statistics, like a training step
    batch = make_batch([0], n_points=2000)
    lidar = batch["lidar"]
    feats, coords = lidar.feats.clone(), lidar.coords.clone()
    captured = {}
    model.classifier.register_forward_hook(lambda m, i, o: captured.__setitem__("logits", o.detach().clone()))
    torch.Tensor.cuda = lambda self, *a, **k: self  # the model calls .cuda() on targets
    ret, _, _ = model(batch)
    return {"feats": feats.numpy(), "coords": coords.numpy(), "labels": batch["targets"].feats.numpy(),
            "logits": captured["logits"].numpy(), "loss": np.array(float(ret["loss"])),
            "state_keys": np.array(sorted(model.state_dict().keys()))}


def _cfg(**kw):
    return _AttrDict(IGNORE_LABEL=0, DROPOUT_P=0.0, IF_DIST=False, **kw)


def cylinder_inputs(seed=0, n_points=2500, grid=(120, 90, 16)):
    """Small cylindrical partition of a synthetic scan, following the reference dataset transform
    (R:pcseg/data/dataset/semantickitti/semantickitti_cylinder.py:19-22,144-160) on a reduced grid."""
    from openpcseg_amd.workloads.synthetic import make_scan
    pts = make_scan(seed, n_points)
    xyz = pts[:, :3]
    rho = np.sqrt(xyz[:, 0] ** 2 + xyz[:, 1] ** 2)
    phi = np.arctan2(xyz[:, 1], xyz[:, 0])
    pol = np.stack([rho, phi, xyz[:, 2]], axis=1)
    lo, hi = np.array([0.0, -np.pi, -4.0]), np.array([50.0, np.pi, 2.0])
    pol = np.clip(pol, lo, hi)
    intervals = (hi - lo) / (np.array(grid) - 1)
    gi = np.floor((pol - lo) / intervals).astype(np.int32)
    centre = (gi.astype(np.float32) + 0.5) * intervals + lo
    feat = np.concatenate([pol - centre, pol, xyz[:, :2], pts[:, 3:4]], axis=1).astype(np.float32)  # 9 dims
    rng = np.random.default_rng(seed + 7)
    plabel = rng.integers(0, 20, size=n_points).astype(np.int64)
    pc = np.concatenate([gi, np.zeros((n_points, 1), np.int32)], axis=1)
    vox, first = np.unique(pc, axis=0, return_index=True)
    return {"point_feature": torch.from_numpy(feat), "point_coord": torch.from_numpy(pc),
            "voxel_coord": torch.from_numpy(vox.astype(np.int32)), "voxel_label": torch.from_numpy(plabel[first]),
            "point_label": torch.from_numpy(plabel), "offset": torch.tensor([n_points], dtype=torch.int32)}


def import_reference_model(dotted):
    import_reference_minkunet()  # installs the import stubs + sys.path
    import importlib
    return importlib.import_module(dotted)


def run_reference_spvcnn():
    from openpcseg_amd.workloads.synthetic import make_batch
    mod = import_reference_model("pcseg.model.segmentor.fusion.spvcnn.spvcnn")
    cfg = _cfg(NAME="SPVCNN", IN_FEATURE_DIM=4, BLOCK="ResBlock", NUM_LAYER=[2, 2, 2, 2, 2, 2, 2, 2],
               PLANES=[32, 32, 64, 128, 256, 256, 128, 96, 96], cr=0.25, LABEL_SMOOTHING=0.1)
    torch.manual_seed(0)
    model = mod.SPVCNN(cfg, 20)
    seeded_state(model)
    model.train()
    batch = make_batch([3], n_points=2000)
    feats, coords = batch["lidar"].feats.clone(), batch["lidar"].coords.clone()
    cap = {}
    model.classifier.register_forward_hook(lambda m, i, o: cap.__setitem__("logits", o.detach().clone()))
    orig = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        ret, _, _ = model(batch)
    finally:
        torch.Tensor.cuda = orig
    return {"spv_feats": feats.numpy(), "spv_coords": coords.numpy(), "spv_labels": batch["targets"].feats.numpy(),
            "spv_logits": cap["logits"].numpy(), "spv_loss": np.array(float(ret["loss"].detach()))}


def run_reference_cylinder():
    mod = import_reference_model("pcseg.model.segmentor.voxel.cylinder3d.cylinder_ts")
    cfg = _cfg(NAME="Cylinder_TS", IN_FEATURE_DIM=9, LABEL_SMOOTHING=0.0, INIT_SIZE=8, POINT_REFINEMENT=True)
    torch.manual_seed(0)
    model = mod.Cylinder_TS(cfg, 20)
    seeded_state(model)
    model.train()
    inp = cylinder_inputs()
    cap = {}
    model.logits.register_forward_hook(lambda m, i, o: cap.__setitem__("logits", (o.F.detach().clone(), o.C.clone())))
    orig = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        ret = model({k: v.clone() for k, v in inp.items()})
    finally:
        torch.Tensor.cuda = orig
    ret = ret[0] if isinstance(ret, tuple) else ret
    out = {"cyl_" + k: v.numpy() for k, v in inp.items()}
    out["cyl_logits"], out["cyl_logit_coords"] = cap["logits"][0].numpy(), cap["logits"][1].numpy()
    out["cyl_loss"] = np.array(float(ret["loss"].detach()))
    return out


def rpvnet_inputs(seed=5, n_points=2000, h=64, w=512):
    """lidar batch + (5,H,W) range image + per-point (batch, px, py) in [-1,1], in the spirit of
    R:pcseg/data/dataset/semantickitti/semantickitti_fusion.py:64-114 (spherical projection)."""
    from openpcseg_amd.workloads.synthetic import make_batch, make_scan
    batch = make_batch([seed], n_points=n_points)
    pts = batch["lidar"].feats.numpy()
    xyz = pts[:, :3]
    depth = np.linalg.norm(xyz, axis=1) + 1e-6
    yaw, pitch = -np.arctan2(xyz[:, 1], xyz[:, 0]), np.arcsin(xyz[:, 2] / depth)
    fu, fd = np.deg2rad(3.0), np.deg2rad(-25.0)
    px = 0.5 * (yaw / np.pi + 1.0)
    py = 1.0 - (pitch - fd) / (fu - fd)
    ix = np.clip(np.floor(px * w), 0, w - 1).astype(np.int64)
    iy = np.clip(np.floor(py * h), 0, h - 1).astype(np.int64)
    img = np.zeros((5, h, w), np.float32)
    img[0, iy, ix], img[1, iy, ix] = depth, pts[:, 3]
    img[2:, iy, ix] = xyz.T
    pxpy = np.stack([np.zeros(len(px)), np.clip(px * 2 - 1, -1, 1), np.clip(py * 2 - 1, -1, 1)], 1).astype(np.float32)
    batch["range_image"] = torch.from_numpy(img)[None]
    batch["range_pxpy"] = torch.from_numpy(pxpy)
    return batch


def run_reference_rpvnet(tag="rpv", in_dim=4, num_class=20, label_smoothing=0.1, seed=5):
    """RPVNet (config 5). range_lib has no CPU build in the reference: its two ops are served here by the
    restatement of RL:range_utils/src/*.cu (oracle.map_count / denselize_fwd); everything else is the reference."""
    fn = install_range_stub()
    mod = import_reference_model("pcseg.model.segmentor.fusion.rpvnet.rpvnet")
    mod.rnf = fn
    # the reference's IF_DIST=False variant is broken (rpvnet.py:574 applies the SparseTensor BatchNorm wrapper
    # to plain tensors), so the shipped IF_DIST=True variant is used -- in TRAIN mode: with no process group
    # initialised nn.SyncBatchNorm computes plain batch statistics (torch/nn/modules/batchnorm.py need_sync), so the
    # logits stay O(10) and an absolute bound means something. The range image is 16 x 128 so that 2000 rays fill it
    # (batch statistics over a mostly empty image blow the activations up).
    cfg = _cfg(NAME="RPVNet", IN_FEATURE_DIM=in_dim, BLOCK="ResBlock", NUM_LAYER=[2] * 8,
               PLANES=[32, 32, 64, 128, 256, 256, 128, 96, 96], cr=0.25, LABEL_SMOOTHING=label_smoothing)
    cfg["IF_DIST"] = True
    torch.manual_seed(0)
    model = mod.RPVNet(cfg, num_class)
    seeded_state(model)
    model.train()
    import fullsize
    fullsize.freeze_dropout(model)  # the range branch's hard-coded Dropout2d(0.2): random masks are not reproducible
    batch = rpvnet_inputs(seed=seed, h=16, w=128)
    if in_dim == 5:  # Waymo's fifth point feature (elongation): a deterministic function of the point, as tests/golden/fullsize.py
        f = batch["lidar"].feats
        batch["lidar"].feats = torch.cat([f, torch.frac(f[:, :1] * 0.37 + f[:, 3:4] * 1.9).abs()], dim=1).contiguous()
    if num_class > 20:  # use the whole label range of the head
        batch["targets"].feats = (batch["targets"].feats + (batch["lidar"].coords[:, 0] % 2) * (num_class - 20)).long() % num_class
    keep = {tag + "_feats": batch["lidar"].feats.numpy().copy(), tag + "_coords": batch["lidar"].coords.numpy().copy(),
            tag + "_labels": batch["targets"].feats.numpy().copy(), tag + "_range_image": batch["range_image"].numpy().copy(),
            tag + "_range_pxpy": batch["range_pxpy"].numpy().copy()}
    cap = {}
    model.classifier.register_forward_hook(lambda m, i, o: cap.__setitem__("logits", o.detach().clone()))
    orig = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        ret, _, _ = model(batch)
    finally:
        torch.Tensor.cuda = orig
    keep[tag + "_logits"] = cap["logits"].numpy()
    keep[tag + "_loss"] = np.array(float(ret["loss"].detach()))
    return keep


def install_scatter_stub():
    """torch_scatter is not installed here: its two functions via torch.scatter_reduce (same semantics)."""
    ts_mod = types.ModuleType("torch_scatter")

    def scatter_max(src, index, dim=0):
        m = int(index.max()) + 1
        out = torch.zeros(m, src.shape[1]).scatter_reduce(0, index[:, None].expand(-1, src.shape[1]), src, "amax",
                                                          include_self=False)
        return out, None

    def scatter_mean(src, index, dim=0):
        m = int(index.max()) + 1
        return torch.zeros(m, src.shape[1]).scatter_reduce(0, index[:, None].expand(-1, src.shape[1]), src, "mean",
                                                           include_self=False)
    ts_mod.scatter_max, ts_mod.scatter_mean = scatter_max, scatter_mean
    sys.modules["torch_scatter"] = ts_mod


def install_range_stub():
    """range_lib has no CPU build in the reference: its ops are served by the restatement of RL:range_utils/src/*.cu
    (oracle.map_count / denselize_fwd / denselize_bwd) -- parity unpinned for exactly these ops."""
    from oracle import oracle as orc

    class _Dense(torch.autograd.Function):
        @staticmethod
        def forward(ctx, feat, cm, pxpy):
            ctx.save_for_backward(cm, pxpy)
            return torch.from_numpy(orc.denselize_fwd(feat.detach().numpy(), cm.numpy(), pxpy.numpy()))

        @staticmethod
        def backward(ctx, gout):
            cm, pxpy = ctx.saved_tensors
            return torch.from_numpy(orc.denselize_bwd(gout.contiguous().numpy(), cm.numpy(), pxpy.numpy())), None, None

    fn = types.ModuleType("range_utils.nn.functional")
    fn.map_count = lambda pxpy, b, h, w: torch.from_numpy(orc.map_count(pxpy.numpy(), b, h, w))
    fn.denselize = lambda feat, cm, pxpy: _Dense.apply(feat, cm, pxpy)
    for name in ("range_utils", "range_utils.nn"):
        sys.modules[name] = types.ModuleType(name)
    sys.modules["range_utils.nn.functional"] = fn
    sys.modules["range_utils.nn"].functional = fn
    sys.modules["range_utils"].nn = sys.modules["range_utils.nn"]
    return fn


def main_models():
    """SPVCNN (config 3) and Cylinder_TS (config 4): the reference's own model code on the reference backend."""
    import_reference_torchsparse()
    install_scatter_stub()
    g = {}
    g.update(run_reference_spvcnn())
    g.update(run_reference_cylinder())
    g.update(run_reference_rpvnet())
    # the Waymo head: 23 classes, 5 point features, no label smoothing (R:tools/cfgs/fusion/waymo/rpvnet_mk18_cr10.yaml:13-23)
    g.update(run_reference_rpvnet(tag="rpw", in_dim=5, num_class=23, label_smoothing=0.0, seed=6))
    np.savez_compressed(os.path.join(OUT, "models_e2e_golden.npz"), **g)
    print("wrote models_e2e_golden.npz:", {k: v.shape for k, v in g.items() if "logits" in k})


def main_quantize():
    """sparse_quantize of the reference on float points / fractional voxel sizes / negative coordinates with
    return_index + return_inverse (the device path's contract)."""
    import_reference_torchsparse()
    from torchsparse.utils.quantize import sparse_quantize
    from openpcseg_amd.workloads.synthetic import make_scan
    g = {}
    cases = {"scan": (make_scan(seed=3, n_points=6000)[:, :3].astype(np.float32), 0.35),
             "aniso": (make_scan(seed=4, n_points=3000)[:, :3].astype(np.float32), (0.1, 0.2, 0.4)),
             "ints": (np.random.default_rng(11).integers(-40, 40, size=(5000, 3)).astype(np.int32), 3)}
    for name, (pts, vs) in cases.items():
        vox, idx, inv = sparse_quantize(pts, vs, return_index=True, return_inverse=True)
        g[name + "_in"], g[name + "_vs"] = pts, np.asarray(vs, dtype=np.float64)
        g[name + "_vox"], g[name + "_idx"], g[name + "_inv"] = vox, idx.astype(np.int64), inv.astype(np.int64)
    np.savez_compressed(os.path.join(OUT, "quantize_golden.npz"), **g)
    print("wrote quantize_golden.npz:", {k: v.shape for k, v in g.items() if k.endswith("_vox")})


def lovasz_cases():
    """(name, probas (n, C) float32, labels (n,) int64, ignore) of the Lovasz-softmax golden: an ignored label inside
    and outside the class range, no ignore, absent classes, saturated probabilities (exact 0 / 1 errors -> ties), two
    valid points among ignored ones."""
    rng = np.random.default_rng(23)
    out = []
    for name, n, nc, ign, hi in (("kitti", 6000, 20, 0, 20), ("ign255", 4000, 17, 255, 15), ("noign", 3000, 20, None, 20),
                                 ("waymo", 5000, 23, 0, 23), ("tiny", 9, 5, 0, 5)):
        z = rng.normal(size=(n, nc)).astype(np.float32) * 3
        p = np.exp(z - z.max(1, keepdims=True))
        p = (p / p.sum(1, keepdims=True)).astype(np.float32)
        lab = rng.integers(0, hi, n).astype(np.int64)
        if name == "kitti":
            lab[lab == 7] = 3                      # an absent class
            sat = rng.random(n) < 0.05             # saturated rows: one-hot probabilities, right and wrong
            p[sat] = 0.0
            p[sat, rng.integers(0, nc, int(sat.sum()))] = 1.0
        if ign not in (None, 0):
            lab[rng.random(n) < 0.1] = ign
        if name == "tiny":
            lab[:] = 0
            lab[4], lab[7] = 3, 2   # (a single valid point breaks the reference: flatten_probas squeezes it to 1-D)
        out.append((name, p, lab, ign))
    return out


def range_sample_inputs():
    """Two cases for `range_to_point`: (B, C, H, W) feature maps + per-point (frame, x, y) rows grouped by frame, x / y mostly inside
    [-1, 1] with some outside (zero padding), some exactly on pixel centres and on the image border."""
    cases = {}
    for name, (seed, b, c, h, w, n) in {"a": (21, 2, 8, 16, 64, 3000), "b": (22, 3, 20, 8, 32, 2500)}.items():
        rng = np.random.default_rng(seed)
        img = rng.normal(size=(b, c, h, w)).astype(np.float32)
        frames = np.sort(rng.integers(0, b, size=n)).astype(np.float32)
        xy = rng.uniform(-1.08, 1.08, size=(n, 2)).astype(np.float32)
        xy[:50] = np.float32(-1) + (np.float32(2) * rng.integers(0, w, size=(50, 1)).astype(np.float32) + 1) / np.float32(w) * np.array([[1, 0]], np.float32) \
            + xy[:50] * np.array([[0, 1]], np.float32)                      # x exactly on pixel centres
        xy[50:60] = np.array([[-1.0, -1.0], [1.0, 1.0], [-1.0, 1.0], [1.0, -1.0], [0.0, 0.0], [1.0, 0.0], [0.0, 1.0], [-1.0, 0.0],
                              [0.0, -1.0], [0.999999, -0.999999]], np.float32)
        cases[name] = (img, np.concatenate([frames[:, None], xy], 1).astype(np.float32), rng.normal(size=(n, c)).astype(np.float32))
    return cases


def main_range_sample():
    """`range_to_point(feature_map, pxpy)` of the reference (R:pcseg/model/segmentor/fusion/rpvnet/rpvnet.py:31-51: grid_sample per
    frame) run on CPU, forward and the gradient w.r.t. the feature map for a fixed output gradient."""
    import_reference_torchsparse()
    install_scatter_stub()
    install_range_stub()
    mod = import_reference_model("pcseg.model.segmentor.fusion.rpvnet.rpvnet")
    g = {}
    for name, (img, pxpy, gout) in range_sample_inputs().items():
        ti = torch.from_numpy(img).requires_grad_(True)
        out = mod.range_to_point(ti, torch.from_numpy(pxpy), "bilinear")
        assert tuple(out.shape) == gout.shape
        out.backward(torch.from_numpy(gout))
        g[name + "_img"], g[name + "_pxpy"], g[name + "_gout"] = img, pxpy, gout
        g[name + "_out"], g[name + "_gimg"] = out.detach().numpy().copy(), ti.grad.numpy().copy()
    np.savez_compressed(os.path.join(OUT, "range_sample_golden.npz"), **g)
    print("wrote range_sample_golden.npz:", {k: v.shape for k, v in g.items() if k.endswith("_out")})


def main_lovasz():
    """lovasz_softmax(probas, labels, ignore) of the reference (tools/utils/common/lovasz_losses.py, imported by path) on
    CPU float32, value and gradient w.r.t. probas."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_lovasz", "/root/reference/tools/utils/common/lovasz_losses.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    g = {}
    for name, p, lab, ign in lovasz_cases():
        tp = torch.from_numpy(p).requires_grad_(True)
        loss = mod.lovasz_softmax(tp, torch.from_numpy(lab), ignore=ign)
        grad, = torch.autograd.grad(loss, tp)
        g[name + "_loss"], g[name + "_grad"] = np.float32(loss.item()), grad.numpy()
    np.savez_compressed(os.path.join(OUT, "lovasz_golden.npz"), **g)
    print("wrote lovasz_golden.npz:", {k: float(v) for k, v in g.items() if k.endswith("_loss")})


def _reference_cpu_setup(per_frame_kernel_hash):
    """The reference's torchsparse on its compiled CPU backend, with the substitutions main_full's docstring lists."""
    ts = import_reference_torchsparse()
    from oracle import oracle as orc
    backend = sys.modules["torchsparse.backend"]

    def devox_bwd(gout, idx, w, n):
        return torch.from_numpy(orc.devoxelize_bwd(gout.contiguous().numpy(), idx.numpy(), w.numpy(), int(n)))
    backend.devoxelize_backward_cpu = devox_bwd
    if per_frame_kernel_hash:
        # multi-frame batch: the reference's CPU kernel_hash reads the batch index of row 0 for every row (hash_cpu.cpp:29;
        # its CUDA twin hash_cuda.cu:42-46 is right) -- called per frame here, which restates the CUDA semantics
        kh_one = backend.kernel_hash_cpu

        def kernel_hash_per_frame(idx, offsets):
            out = torch.empty((offsets.shape[0], idx.shape[0]), dtype=torch.long)
            for b in idx[:, 3].unique().tolist():
                sel = (idx[:, 3] == b).nonzero().squeeze(1)
                out[:, sel] = kh_one(idx[sel].contiguous(), offsets)
            return out
        backend.kernel_hash_cpu = kernel_hash_per_frame
    install_scatter_stub()
    return ts, install_range_stub()


def main_full(cfg_name):
    """BASELINE configs 2-5 at full size (tests/golden/fullsize.py holds the shared definitions): the reference's own
    segmentor, fp32, ONE full synthetic frame (120 000 rays), TRAIN mode (batch statistics), forward + loss + backward
    (R:train.py:360-371) on the reference's torchsparse + its compiled CPU backend. Inputs are regenerated from the
    seed by the tests, so the fixture keeps their CRCs, every 16th row of the logits, float64 column sums over all rows,
    the loss and per-parameter gradient fingerprints.
    Substitutions, none of which changes semantics: devoxelize_backward_cpu (wrong in the reference's CPU twin,
    devoxelize_cpu.cpp:51-53) -> the restatement of devoxelize_cuda.cu:37-57; torch_scatter / range_lib (no CPU build)
    -> install_scatter_stub / install_range_stub (configs 4 / 5 inherit 'parity unpinned' for those two ops only)."""
    import time
    import fullsize as fs
    ts, rnf = _reference_cpu_setup(per_frame_kernel_hash=cfg_name in fs.BATCH_SEEDS)
    dotted, cls = fs.MODEL_PATH[cfg_name]
    mod = import_reference_model(dotted)
    if cfg_name == "config5":
        mod.rnf = rnf
    n_points = int(os.environ["PCS_GOLDEN_POINTS"]) if os.environ.get("PCS_GOLDEN_POINTS") else None
    torch.manual_seed(0)
    torch.set_num_threads(8)
    model = getattr(mod, cls)(_AttrDict(fs.MODEL_CFG[cfg_name]), 20)
    seeded_state(model)
    model.train()
    fs.freeze_dropout(model)
    batch = fs.build_inputs(cfg_name, ts.SparseTensor, n_points)
    g = fs.input_crcs(cfg_name, batch)
    orig = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    t0 = time.time()
    try:
        logits, loss = fs.run_train_step(cfg_name, model, batch)
    finally:
        torch.Tensor.cuda = orig
    print("%s reference forward + backward: %.1f s" % (cfg_name, time.time() - t0))
    g.update(fs.logits_fingerprint(logits, loss))
    g.update(fs.grad_fingerprint(fs.model_grads(model)))
    g["n_points"] = np.array(-1 if n_points is None else n_points)
    np.savez_compressed(os.path.join(OUT, "%s_golden.npz" % cfg_name), **g)
    print("wrote %s_golden.npz: rows" % cfg_name, logits.shape, "kept", g["logits_rows"].shape, "loss", loss,
          "|logit| max %.2f" % np.abs(logits).max(), "params with grad", len(g["grad_names"]))


def main_trajectory():
    """Stand-in for the mIoU clause of north_star (no dataset / checkpoint here): a TRAINING TRAJECTORY of the reference.
    MinkUNet-18 cr0.5 (tests/golden/fullsize.py MODEL_CFG["trajectory"]) on a two-frame batch of 20 000-ray scans, `steps`
    iterations in the order of R:train.py:355-371 -- zero_grad, forward, loss.backward, clip_grad_norm_, optimizer.step -- with the
    shipped optimizer (SGD momentum 0.9, weight decay 1e-4, clip 10; minkunet_mk34_cr10.yaml:25-33) at a fixed learning rate, on
    the reference's torchsparse + compiled CPU backend. Keeps the loss of every step and a fingerprint (float64 sum / abs-sum /
    abs-max + eight samples) of every parameter and BatchNorm buffer after the last step.
    The run is repeated as TWINS: the same reference, same weights, input features multiplied by (1 + eps N(0, 1)) with
    eps = 1e-6 (three seeds) -- the size of the difference between two correct fp32 implementations of one forward pass (the full-size
    fixtures: logits agree to 1.4e-6 of their scale between the reference's CPU code and the MFMA kernels). How far the twins drift
    from the first run and from each other is how reproducible the reference's OWN trajectory is at that level (the loss has kinks:
    ReLU gates, the Lovasz sort order, and every step amplifies), i.e. the floor for any other summation order -- the test bounds
    follow it where it exceeds the nominal 1e-4 / 1e-3. `make_golden.py trajectory twins` adds twins to an existing fixture."""
    import time
    import fullsize as fs
    from torch.nn.utils import clip_grad_norm_
    ts, _ = _reference_cpu_setup(per_frame_kernel_hash=True)
    from openpcseg_amd.workloads.synthetic import make_batch
    T = fs.TRAJ
    dotted, cls = fs.MODEL_PATH["trajectory"]
    mod = import_reference_model(dotted)
    torch.set_num_threads(8)
    b = make_batch(T["seeds"], n_points=T["n_points"])
    feats, coords, labels = b["lidar"].feats, b["lidar"].coords, b["targets"].feats
    g = {"crc_feats": np.array(fs.crc(feats.numpy())), "crc_coords": np.array(fs.crc(coords.numpy())),
         "crc_labels": np.array(fs.crc(labels.numpy()))}

    def run(f, tag):
        torch.manual_seed(0)
        model = getattr(mod, cls)(_AttrDict(fs.MODEL_CFG["trajectory"]), 20)
        seeded_state(model)
        model.train()
        opt = torch.optim.SGD(model.parameters(), lr=T["lr"], momentum=T["momentum"], weight_decay=T["weight_decay"])
        losses, norms = [], []
        t0 = time.time()
        for it in range(T["steps"]):
            model.train()
            opt.zero_grad()
            batch = {"lidar": ts.SparseTensor(f.clone(), coords), "targets": ts.SparseTensor(labels, coords), "offset": None}
            ret = model(batch)
            loss = ret[0]["loss"].mean()
            loss.backward()
            norms.append(float(clip_grad_norm_(model.parameters(), T["clip"])))
            opt.step()
            losses.append(float(loss.detach()))
            print("%s step %d loss %.6f grad norm %.4f (%.0f s)" % (tag, it, losses[-1], norms[-1], time.time() - t0), flush=True)
        state = [(n, t) for n, t in model.state_dict().items() if t.dtype.is_floating_point]
        return np.array(losses, dtype=np.float64), np.array(norms, dtype=np.float64), fs.grad_fingerprint(state)

    path = os.path.join(OUT, "trajectory_golden.npz")
    twins_only = len(sys.argv) > 2 and sys.argv[2] == "twins" and os.path.exists(path)
    orig = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    if len(sys.argv) > 2 and sys.argv[2] == "threads":
        # `make_golden.py trajectory threads`: the SAME reference run on the SAME inputs, only torch's intra-op thread count changed
        # (8 in the fixture): what the reference's own GEMM blocking / OpenMP reduction order does to its trajectory. Nothing is
        # written to the fixture; the drift per step goes to profiles/round5_trajectory_thread_drift.json.
        import json
        old = np.load(path)
        assert all(int(old[k]) == int(g[k]) for k in g), "the existing fixture was made from other inputs"
        rec = {"about": "reference trajectory (make_golden.py trajectory) re-run with other intra-op thread counts, same inputs, same "
                        "weights, no perturbation; drift = |loss / fixture loss - 1| per step, weights = abs-sum per tensor, relative",
               "fixture_threads": 8, "fixture_losses": [float(v) for v in old["losses"]], "runs": {}}
        try:
            for nt in (1, 3):
                torch.set_num_threads(nt)
                l, n_, fp = run(feats, "threads%d" % nt)
                rec["runs"][str(nt)] = {
                    "losses": [float(v) for v in l], "loss_rel_drift": [float("%.3g" % v) for v in np.abs(l / old["losses"] - 1)],
                    "grad_norm_rel_drift_max": float(np.abs(n_ / old["grad_norms"] - 1).max()),
                    "weights_abssum_rel_drift_max": float(np.abs(fp["grad_stats"][:, 1] / old["state_stats"][:, 1] - 1).max())}
        finally:
            torch.Tensor.cuda = orig
        with open(os.path.join(os.path.dirname(os.path.dirname(OUT)), "profiles", "round5_trajectory_thread_drift.json"), "w") as f:
            json.dump(rec, f, indent=1)
        print(json.dumps(rec["runs"], indent=1))
        return
    try:
        if twins_only:
            old = dict(np.load(path))
            assert all(int(old[k]) == int(g[k]) for k in g), "the existing fixture was made from other inputs"
            g = {k: v for k, v in old.items() if not k.startswith("twin")}
            losses = g["losses"]
        else:
            losses, norms, fp = run(feats, "main")
            g["losses"], g["grad_norms"] = losses, norms
            g["state_names"], g["state_stats"], g["state_samples"] = fp["grad_names"], fp["grad_stats"], fp["grad_samples"]
        eps, tl, ts_, tm = 1e-6, [], [], []
        for seed in (7, 8, 9):
            noise = torch.randn(feats.shape, generator=torch.Generator().manual_seed(seed))
            t_losses, _, t_fp = run(feats * (1.0 + eps * noise), "twin%d" % seed)
            tl.append(t_losses); ts_.append(t_fp["grad_stats"]); tm.append(t_fp["grad_samples"])
    finally:
        torch.Tensor.cuda = orig
    g["twin_eps"], g["twin_losses"] = np.array(eps), np.array(tl)                       # (3, steps)
    g["twin_state_stats"], g["twin_state_samples"] = np.array(ts_), np.array(tm)        # (3, tensors, 3), (3, tensors, 8)
    np.savez_compressed(path, **g)
    print("wrote trajectory_golden.npz: losses", np.asarray(losses).tolist(), "\ntwin drift", np.abs(np.array(tl) / losses - 1).max(0).tolist())


def main_cylinder():
    """Cylinder front-end: the reference's OWN dataset transform
    (R:pcseg/data/dataset/semantickitti/semantickitti_cylinder.py:19-45 cart2polar / voxelize_with_label and
    :144-173 get_single_sample, inference branch = no augmentation) run on synthetic scans with the shipped
    cy480 configuration (R:tools/cfgs/voxel/semantic_kitti/cylinder_cy480_cr10.yaml:7-9) and a reduced grid.
    The class is instantiated without its __init__ (which wants the dataset on disk); `np.int`, removed from
    NumPy >= 1.24 and still used by the reference (:158,:170-175), is aliased to `int` for the run."""
    import_reference_torchsparse()
    import_reference_minkunet()  # import stubs + sys.path for the reference tree
    import importlib
    if not hasattr(np, "int"):
        np.int = int
    mod = importlib.import_module("pcseg.data.dataset.semantickitti.semantickitti_cylinder")
    from openpcseg_amd.workloads.synthetic import make_scan
    g = {}
    cases = {"cy480": (6, 20000, [0, -180, -4], [50, 180, 2], [480, 360, 32]),
             "small": (7, 5000, [0, -180, -4], [50, 180, 2], [120, 90, 16]),
             "clip": (8, 4000, [2, -90, -2], [30, 120, 1], [64, 48, 8])}   # points outside the cylinder are clipped
    for name, (seed, n, lo, hi, grid) in cases.items():
        pts = make_scan(seed, n).astype(np.float32)
        rng = np.random.default_rng(seed + 100)
        labels = rng.integers(0, 20, size=n).astype(np.int64)
        labels[rng.random(n) < 0.05] = 67   # the "don't count" label of voxelize_with_label (:36)
        ds = object.__new__(mod.SemkittiCylinderDataset)
        ds.training, ds.if_tta = False, False
        ds.class_names = ["c%d" % i for i in range(20)]
        ds.cylinder_space_max, ds.cylinder_space_min, ds.grid_size = np.array(hi), np.array(lo), np.array(grid)
        ds.point_cloud_dataset = [{"labels": labels.copy(), "xyzret": pts.copy(), "path": name}]
        ret = ds.get_single_sample(0)
        g[name + "_points"], g[name + "_labels"] = pts, labels
        g[name + "_cfg"] = np.array([lo, hi, grid], dtype=np.int64)
        for k in ("point_feature", "point_coord", "voxel_feature", "voxel_coord", "voxel_label", "inverse_map"):
            g[name + "_" + k] = ret[k]
    np.savez_compressed(os.path.join(OUT, "cylinder_golden.npz"), **g)
    print("wrote cylinder_golden.npz:", {k: v.shape for k, v in g.items() if k.endswith("voxel_coord")})


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "cylinder":
        main_cylinder()
    elif len(sys.argv) > 1 and sys.argv[1] in ("config2", "config3", "config4", "config5", "config2x2", "config_mk34"):
        main_full(sys.argv[1])
    elif len(sys.argv) > 1 and sys.argv[1] == "trajectory":
        main_trajectory()
    elif len(sys.argv) > 1 and sys.argv[1] == "range_sample":
        main_range_sample()
    elif len(sys.argv) > 1 and sys.argv[1] == "models":
        main_models()
    elif len(sys.argv) > 1 and sys.argv[1] == "quantize":
        main_quantize()
    elif len(sys.argv) > 1 and sys.argv[1] == "lovasz":
        main_lovasz()
    else:
        main()
