"""BASELINE configs 2-5 at (near) full size: what make_golden.py [config2|config3|config4|config5] runs through the
REFERENCE and what tests/test_fullsize_parity.py re-runs through libpcseg_hip.so -- one definition of the inputs, the
model configurations and the fingerprints, imported by both sides.

Inputs are regenerated from seeds (synthetic scans), never stored; the fixtures keep CRCs of every input array, so a
test that does not rebuild the frame the reference saw fails before it compares anything. A fixture keeps, of the
reference run: every `row_step`-th row of the logits, float64 column sums / absolute sums over ALL rows, the loss, and
-- after `loss.backward()` (R:train.py:360-371) -- per-parameter gradient fingerprints: float64 sum, absolute sum,
absolute maximum and eight sampled elements of every `.grad`.
"""
import zlib

import numpy as np
import torch

ROW_STEP = 16
N_GRAD_SAMPLES = 8

MODEL_CFG = {
    # R:tools/cfgs/voxel/semantic_kitti/minkunet_mk18_cr10.yaml
    "config2": dict(NAME="MinkUNet", IGNORE_LABEL=0, IN_FEATURE_DIM=4, BLOCK="ResBlock", NUM_LAYER=[2] * 8,
                    PLANES=[32, 32, 64, 128, 256, 256, 128, 96, 96], cr=1.0, DROPOUT_P=0.0, LABEL_SMOOTHING=0.1,
                    IF_DIST=False),
    # R:tools/cfgs/fusion/semantic_kitti/spvcnn_mk18_cr10.yaml (IF_DIST False: one rank, plain BatchNorm)
    "config3": dict(NAME="SPVCNN", IGNORE_LABEL=0, IN_FEATURE_DIM=4, BLOCK="ResBlock", NUM_LAYER=[2] * 8,
                    PLANES=[32, 32, 64, 128, 256, 256, 128, 96, 96], cr=1.0, DROPOUT_P=0.0, LABEL_SMOOTHING=0.1,
                    IF_DIST=False),
    # R:tools/cfgs/voxel/semantic_kitti/cylinder_cy480_cr10.yaml
    "config4": dict(NAME="Cylinder_TS", IGNORE_LABEL=0, IN_FEATURE_DIM=9, DROPOUT_P=0.0, LABEL_SMOOTHING=0.0,
                    INIT_SIZE=32, POINT_REFINEMENT=True, IF_DIST=False),
    # R:tools/cfgs/fusion/semantic_kitti/rpvnet_mk34_cr17_5.yaml. IF_DIST=True is the shipped variant (the
    # IF_DIST=False RPVNet is broken, rpvnet.py:574); with no process group initialised nn.SyncBatchNorm computes
    # plain batch statistics, so TRAIN mode runs on one rank and the logits stay O(10).
    "config5": dict(NAME="RPVNet", IGNORE_LABEL=0, IN_FEATURE_DIM=5, BLOCK="ResBlock", NUM_LAYER=[2, 3, 4, 6, 2, 2, 2, 2],
                    PLANES=[32, 32, 64, 128, 256, 256, 128, 96, 96], cr=1.75, DROPOUT_P=0.0, LABEL_SMOOTHING=0.0,
                    IF_DIST=True),
}
# config 2 on a TWO-frame batch (seeds 0 and 4): batch-level BatchNorm statistics, loss and gradients over frames that never
# share a voxel (the batch index is part of every hashed coordinate) -- what the one-frame fixtures cannot pin
MODEL_CFG["config2x2"] = MODEL_CFG["config2"]
# the graph BASELINE.json's metric is quoted on (and bench.py's headline times): MinkUNet-34 cr1.0,
# R:tools/cfgs/voxel/semantic_kitti/minkunet_mk34_cr10.yaml:17-20 (NUM_LAYER [2,3,4,6,2,2,2,2]); one full frame, seed 6
MODEL_CFG["config_mk34"] = dict(MODEL_CFG["config2"], NUM_LAYER=[2, 3, 4, 6, 2, 2, 2, 2])
# training-trajectory fixture (make_golden.py trajectory): MinkUNet-18 cr0.5, the optimizer of the shipped yaml
# (R:tools/cfgs/voxel/semantic_kitti/minkunet_mk34_cr10.yaml:25-33: SGD, momentum 0.9, weight decay 1e-4, clip 10) at a fixed lr
MODEL_CFG["trajectory"] = dict(MODEL_CFG["config2"], cr=0.5)
TRAJ = dict(seeds=[11, 12], n_points=20000, steps=10, lr=0.002, momentum=0.9, weight_decay=1e-4, clip=10.0)
BATCH_SEEDS = {"config2x2": [0, 4]}
MODEL_PATH = {"config2x2": ("pcseg.model.segmentor.voxel.minkunet.minkunet", "MinkUNet"),
              "config_mk34": ("pcseg.model.segmentor.voxel.minkunet.minkunet", "MinkUNet"),
              "trajectory": ("pcseg.model.segmentor.voxel.minkunet.minkunet", "MinkUNet"),
              "config2": ("pcseg.model.segmentor.voxel.minkunet.minkunet", "MinkUNet"),
              "config3": ("pcseg.model.segmentor.fusion.spvcnn.spvcnn", "SPVCNN"),
              "config4": ("pcseg.model.segmentor.voxel.cylinder3d.cylinder_ts", "Cylinder_TS"),
              "config5": ("pcseg.model.segmentor.fusion.rpvnet.rpvnet", "RPVNet")}
FRAME_SEED = {"config2": 0, "config3": 1, "config4": 2, "config5": 3, "config_mk34": 6}
CYL_LO, CYL_HI, CYL_GRID = [0, -180, -4], [50, 180, 2], [480, 360, 32]   # cylinder_cy480_cr10.yaml:7-9
RANGE_H, RANGE_W = 64, 2048                                                # SemanticKITTI range image of the reference


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes())


def _t(a):
    return a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))


# ---- inputs ---------------------------------------------------------------------------------------------------------
def lidar_frame(seed, SparseTensor, n_points=None, elongation=False):
    """One voxelised synthetic frame as the reference's voxel dataset hands it over (host tensors)."""
    from openpcseg_amd.workloads.synthetic import make_batch
    b = make_batch([seed], n_points=n_points)
    feats, coords, labels = b["lidar"].feats, b["lidar"].coords, b["targets"].feats
    if elongation:  # Waymo's fifth channel (IN_FEATURE_DIM 5): a deterministic function of the point
        e = torch.frac(feats[:, :1] * 0.37 + feats[:, 3:4] * 1.9).abs()
        feats = torch.cat([feats, e], dim=1).contiguous()
    return {"lidar": SparseTensor(feats, coords), "targets": SparseTensor(labels, coords), "offset": None}


def range_view(feats, h=RANGE_H, w=RANGE_W):
    """(1,5,H,W) range image + per-point (batch, px, py) in [-1,1]: spherical projection in the spirit of
    R:pcseg/data/dataset/semantickitti/semantickitti_fusion.py:64-114 (plain NumPy, deterministic)."""
    pts = feats.numpy() if isinstance(feats, torch.Tensor) else feats
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
    return torch.from_numpy(img)[None], torch.from_numpy(pxpy)


def cylinder_frame(seed, n_points=None):
    """One scan through the reference's OWN dataset transform (SemkittiCylinderDataset.get_single_sample inference
    branch + collate_batch, R:pcseg/data/dataset/semantickitti/semantickitti_cylinder.py:99-213) on the shipped cy480
    grid. The dataset module is pure NumPy and is staged to the GPU box (tests/golden/stage_reference.py EXTRA)."""
    import importlib
    from openpcseg_amd.workloads.synthetic import make_scan
    if not hasattr(np, "int"):
        np.int = int  # removed from NumPy >= 1.24, still used by the reference
    mod = importlib.import_module("pcseg.data.dataset.semantickitti.semantickitti_cylinder")
    pts = make_scan(seed, n_points).astype(np.float32)
    rng = np.random.default_rng(seed + 100)
    labels = rng.integers(0, 20, size=pts.shape[0]).astype(np.int64)
    ds = object.__new__(mod.SemkittiCylinderDataset)
    ds.training, ds.if_tta = False, False
    ds.class_names = ["c%d" % i for i in range(20)]
    ds.cylinder_space_max, ds.cylinder_space_min, ds.grid_size = np.array(CYL_HI), np.array(CYL_LO), np.array(CYL_GRID)
    ds.point_cloud_dataset = [{"labels": labels, "xyzret": pts, "path": "synthetic%d" % seed}]
    b = mod.SemkittiCylinderDataset.collate_batch([ds.get_single_sample(0)])
    return {k: b[k] for k in ("point_feature", "point_coord", "voxel_coord", "voxel_label", "point_label", "offset")}


def build_inputs(cfg_name, SparseTensor, n_points=None):
    if cfg_name in BATCH_SEEDS:
        from openpcseg_amd.workloads.synthetic import make_batch
        b = make_batch(BATCH_SEEDS[cfg_name], n_points=n_points)
        return {"lidar": SparseTensor(b["lidar"].feats, b["lidar"].coords), "targets": SparseTensor(b["targets"].feats, b["lidar"].coords),
                "offset": None}
    seed = FRAME_SEED[cfg_name]
    if cfg_name == "config4":
        return cylinder_frame(seed, n_points)
    b = lidar_frame(seed, SparseTensor, n_points, elongation=(cfg_name == "config5"))
    if cfg_name == "config5":
        b["range_image"], b["range_pxpy"] = range_view(b["lidar"].feats)
    return b


def input_crcs(cfg_name, batch):
    if cfg_name == "config4":
        return {"crc_" + k: np.array(crc(batch[k].numpy())) for k in sorted(batch)}
    out = {"crc_feats": np.array(crc(batch["lidar"].feats.numpy())), "crc_coords": np.array(crc(batch["lidar"].coords.numpy())),
           "crc_labels": np.array(crc(batch["targets"].feats.numpy()))}
    if cfg_name == "config5":
        out["crc_range_image"] = np.array(crc(batch["range_image"].numpy()))
        out["crc_range_pxpy"] = np.array(crc(batch["range_pxpy"].numpy()))
    return out


def to_device(cfg_name, batch, dev, SparseTensor):
    if cfg_name == "config4":
        return {k: v.to(dev) for k, v in batch.items()}
    c = batch["lidar"].coords.to(dev)
    out = {"lidar": SparseTensor(batch["lidar"].feats.to(dev), c), "targets": SparseTensor(batch["targets"].feats.to(dev), c),
           "offset": None}
    for k in ("range_image", "range_pxpy"):
        if k in batch:
            out[k] = batch[k].to(dev)
    return out


# ---- running a reference segmentor ------------------------------------------------------------------------------------
def logits_hook(cfg_name, model, cap, via="classifier"):
    """The tensor the 1e-3 bound of north_star is about: per-point (config 4: per-voxel) logits before the loss.
    via="criterion": taken as the first argument of the criterion instead of the classifier's output -- the same tensor in the
    reference's forward (`out = self.classifier(..); loss = self.criterion_losses(out, ..)`), and the one that exists when the
    fused forward forms the class scores on the voxels (openpcseg_amd/block_fusion.py)."""
    if via == "criterion" and cfg_name != "config4":
        return model.criterion_losses.register_forward_pre_hook(lambda m, args: cap.__setitem__("logits", args[0].detach()))
    if cfg_name == "config4":
        return model.logits.register_forward_hook(lambda m, i, o: cap.__setitem__("logits", o.F.detach()))
    return model.classifier.register_forward_hook(lambda m, i, o: cap.__setitem__("logits", o.detach()))


def freeze_dropout(model):
    """Train mode everywhere EXCEPT the dropout layers: RPVNet's range branch hard-codes Dropout2d(p=0.2)
    (rpvnet.py:211-222), whose random masks cannot be reproduced across devices / generators -- they are switched to
    identity on both sides of the comparison. Call after model.train()."""
    for m in model.modules():
        if isinstance(m, (torch.nn.Dropout, torch.nn.Dropout2d, torch.nn.Dropout3d)):
            m.eval()
    return model


def run_train_step(cfg_name, model, batch, via="classifier"):
    """forward (train mode, batch statistics) + loss + backward; returns (logits ndarray, loss float)."""
    cap = {}
    h = logits_hook(cfg_name, model, cap, via)
    try:
        ret = model(batch)
    finally:
        h.remove()
    ret = ret[0] if isinstance(ret, tuple) else ret
    loss = ret["loss"]
    model.zero_grad(set_to_none=True)
    loss.backward()
    return cap["logits"].detach().float().cpu().numpy(), float(loss.detach())


# ---- fingerprints -----------------------------------------------------------------------------------------------------
def grad_fingerprint(named_grads):
    """named_grads: iterable of (name, grad tensor). float64 [sum, abs-sum, abs-max] + N_GRAD_SAMPLES evenly spaced elements."""
    names, stats, samples = [], [], []
    for name, g in sorted(named_grads, key=lambda kv: kv[0]):
        g = g.detach().double().cpu().reshape(-1)
        idx = torch.from_numpy(np.linspace(0, g.numel() - 1, N_GRAD_SAMPLES).astype(np.int64))
        names.append(name)
        stats.append([float(g.sum()), float(g.abs().sum()), float(g.abs().max())])
        samples.append(g[idx].numpy())
    return {"grad_names": np.array(names), "grad_stats": np.array(stats, dtype=np.float64),
            "grad_samples": np.array(samples, dtype=np.float64)}


def model_grads(model):
    return [(n, p.grad) for n, p in model.named_parameters() if p.grad is not None]


def logits_fingerprint(logits, loss):
    l64 = logits.astype(np.float64)
    return {"n_rows": np.array(logits.shape[0]), "row_step": np.array(ROW_STEP), "logits_rows": logits[::ROW_STEP].copy(),
            "logits_colsum": l64.sum(0), "logits_abssum": np.abs(l64).sum(0), "loss": np.array(loss)}


def compare(g, logits, loss, named_grads=None):
    """Measured errors of a run against a fixture (a dict of plain floats; the caller asserts its bounds on them)."""
    step = int(g["row_step"])
    ref = g["logits_rows"]
    out = {"logit_scale": float(np.abs(ref).max()), "logit_rms": float(np.sqrt((ref.astype(np.float64) ** 2).mean())),
           "logit_max_abs_err": float(np.abs(logits[::step] - ref).max()),
           "colsum_err_per_row": float(np.abs(logits.astype(np.float64).sum(0) - g["logits_colsum"]).max() / logits.shape[0]),
           "abssum_rel_err": float(np.abs(np.abs(logits.astype(np.float64)).sum(0) / g["logits_abssum"] - 1).max()),
           "loss_abs_err": abs(loss - float(g["loss"]))}
    if named_grads is not None and "grad_names" in g:
        mine = grad_fingerprint(named_grads)
        ref_names = [str(n) for n in g["grad_names"]]
        assert [str(n) for n in mine["grad_names"]] == ref_names, "parameter sets differ"
        rs, ms = g["grad_stats"], mine["grad_stats"]
        # Gradients that are ZERO by construction (a bias in front of a train-mode BatchNorm: the normalisation removes
        # it) come out as rounding noise on both sides; they are "dead": checked to stay noise, excluded from the relative
        # errors. G = the typical gradient size of the model (median over parameters of the largest |element|).
        G = float(np.median(rs[:, 2]))
        live = rs[:, 2] > 1e-3 * G   # measured gap: dead <= 2.4e-4 G, live >= 3.4e-2 G (config 4)
        out["grad_params"], out["grad_dead_params"] = int(len(ref_names)), int((~live).sum())
        out["grad_dead_max_over_G"] = float(ms[~live, 2].max() / G) if (~live).any() else 0.0
        # per live parameter, relative to that gradient's own size: abs-sum (no cancellation inside it), the signed sum
        # against the abs-sum, eight sampled elements against the largest element
        e_abs = np.abs(ms[live, 1] / rs[live, 1] - 1)
        e_sum = np.abs(ms[live, 0] - rs[live, 0]) / rs[live, 1]
        e_smp = np.abs(mine["grad_samples"][live] - g["grad_samples"][live]).max(1) / rs[live, 2]
        names = [n for n, l in zip(ref_names, live) if l]
        for key, e in (("grad_abssum_rel_err", e_abs), ("grad_sum_err_rel_abssum", e_sum), ("grad_sample_err_rel_max", e_smp)):
            out[key] = float(e.max())
            out[key + "_median"] = float(np.median(e))
            out[key + "_p90"] = float(np.percentile(e, 90))
        out["grad_worst_param"] = names[int(np.argmax(e_abs))]
        # the big tensors (>= 2-D weights: convolution and linear kernels) on their own: sums over many rows, no
        # cancellation to speak of -- the tight bound of the test
        shapes = {n: tuple(t.shape) for n, t in named_grads}
        mat = np.array([len(shapes[n]) >= 2 for n in names])
        if mat.any():
            out["grad_matrix_params"] = int(mat.sum())
            out["grad_matrix_abssum_rel_err"] = float(e_abs[mat].max())
            out["grad_matrix_sample_err_rel_max"] = float(e_smp[mat].max())
            out["grad_matrix_worst_param"] = [n for n, m_ in zip(names, mat) if m_][int(np.argmax(e_abs[mat]))]
    return out
