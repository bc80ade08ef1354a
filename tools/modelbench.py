#!/usr/bin/env python3
"""Throughput of the route INTEGRATION.md section 1 describes: the reference's OWN segmentor sources, imported unmodified
on top of `openpcseg_amd.install_reference_aliases()`, training on the HIP backend -- next to this package's fused
MinkUNet workload. One record per (model, source, dtype) for BASELINE configs 2-5 at their BATCH_SIZE_PER_GPU:

  minkunet18 (config 2, 16 frames), spvcnn18 (config 3, 16), cylinder (config 4: Cylinder_TS cy480, 12), rpvnet34 (config 5:
  RPVNet mk34 cr 1.75 with the 5-channel input and a (5, 64, 2048) range image per frame, 4), minkunet34 (the headline model, 12)

  R:pcseg/model/segmentor/voxel/minkunet/minkunet.py:385-434, fusion/spvcnn/spvcnn.py, voxel/cylinder3d/cylinder_ts.py,
  fusion/rpvnet/rpvnet.py; batch sizes from R:tools/cfgs/{voxel,fusion}/semantic_kitti/*.yaml.

A step = zero_grad + forward + loss + backward + SGD step on synthetic 120k-point scans resident in HBM (every kernel map
rebuilt each step, as in the reference). The reference sources are read from /root/reference or their staged copy
tests/_refsrc (bench plumbing: nothing under openpcseg_amd/ imports this file).

    python tools/modelbench.py [spec]        spec = "auto" | comma list of name[:reference|fuse|workload][:f32|bf16]

source `fuse` = the reference's source after `openpcseg_amd.fuse(model)` (block fusion, openpcseg_amd/block_fusion.py): the record
`<name>/ref+fuse`. Timing [r5]: clock pre-heat (windows of 5 steps until a window is no longer > 0.4 % faster than the one before,
2..3 windows) and PCS_MB_STEPS (default 10) timed steps, like the headline.
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

FRAMES = {"minkunet18": 16, "spvcnn18": 16, "cylinder": 12, "rpvnet34": 4, "minkunet34": 12}
CFG_OF = {"minkunet18": "config2", "spvcnn18": "config3", "cylinder": "config4", "rpvnet34": "config5", "minkunet34": "config2"}
AUTO = ["minkunet34:reference", "minkunet34:fuse", "minkunet18:reference", "minkunet18:fuse", "minkunet18:workload",
        "spvcnn18:reference", "spvcnn18:fuse", "cylinder:reference", "cylinder:fuse", "rpvnet34:reference", "rpvnet34:fuse"]


def _reference_model(name):
    import fullsize as fs
    import make_golden as mg
    import openpcseg_amd
    from seeded import seeded_state
    openpcseg_amd.install_reference_aliases()
    cfg_name = CFG_OF[name]
    dotted, cls = fs.MODEL_PATH[cfg_name]
    mod = mg.import_reference_model(dotted)
    for m in list(sys.modules.values()):
        if getattr(m, "__name__", "").startswith(("pcseg.", "tools.")) and hasattr(m, "torch_scatter"):
            m.torch_scatter = sys.modules["torch_scatter"]
    if cfg_name == "config5":
        mod.rnf = sys.modules["range_utils.nn.functional"]
    cfg = dict(fs.MODEL_CFG[cfg_name])
    if name == "minkunet34":
        cfg["NUM_LAYER"] = [2, 3, 4, 6, 2, 2, 2, 2]  # R:tools/cfgs/voxel/semantic_kitti/minkunet_mk34_cr10.yaml:17-20
    model = getattr(mod, cls)(mg._AttrDict(cfg), 20)
    seeded_state(model)
    return model


_BATCHES = {}


def _lidar_batch(n_frames, dev, elongation=False, range_view=False):
    key = (n_frames, str(dev), elongation, range_view)
    if key not in _BATCHES:   # the same synthetic batch serves every record of one run (generated on the host: seconds each)
        _BATCHES.clear()
        _BATCHES[key] = _make_lidar_batch(n_frames, dev, elongation, range_view)
    return _BATCHES[key]


def _make_lidar_batch(n_frames, dev, elongation=False, range_view=False):
    import fullsize as fs
    from openpcseg_amd.sparse import SparseTensor
    from openpcseg_amd.workloads.synthetic import make_batch
    npts = int(os.environ["PCS_MB_POINTS"]) if os.environ.get("PCS_MB_POINTS") else None   # host-overhead runs: tiny scans
    b = make_batch(list(range(n_frames)), n_points=npts) if npts else make_batch(list(range(n_frames)))
    feats, coords, labels = b["lidar"].feats, b["lidar"].coords, b["targets"].feats
    out = {}
    if elongation:
        e = torch.frac(feats[:, :1] * 0.37 + feats[:, 3:4] * 1.9).abs()
        feats = torch.cat([feats, e], dim=1).contiguous()
    if range_view:  # one (5, H, W) image per frame, per-point (batch, px, py)
        imgs, pxpy = [], []
        for i in range(n_frames):
            sel = coords[:, 3] == i
            img, pp = fs.range_view(feats[sel])
            pp[:, 0] = i
            imgs.append(img)
            pxpy.append(pp)
        out["range_image"] = torch.cat(imgs, 0).to(dev)
        out["range_pxpy"] = torch.cat(pxpy, 0).to(dev)
    c = coords.to(dev)
    f, l = feats.to(dev), labels.to(dev)

    def fresh():
        # NEW tensor objects every step (views of the resident data): the memos this package hangs on caller tensors (pixel / corner
        # CSRs, integer pxpy, frame-order flags: native._cached) must be rebuilt every step like in real training, where every
        # batch is a new tensor -- round 5 handed the same objects to every step and never paid them after the first
        cc = c.view_as(c)
        d = {"lidar": SparseTensor(f.view_as(f), cc), "targets": SparseTensor(l.view_as(l), cc), "offset": None}
        d.update({k: v.view_as(v) for k, v in out.items()})
        return d
    return fresh, int(c.shape[0])


def _cylinder_batch(n_frames, dev):
    """n_frames scans through the reference's own dataset transform + collate (as tests/golden/fullsize.py::cylinder_frame)."""
    import importlib
    import fullsize as fs
    import make_golden
    from openpcseg_amd.workloads.synthetic import make_scan
    make_golden.import_reference_minkunet()
    if not hasattr(np, "int"):
        np.int = int
    mod = importlib.import_module("pcseg.data.dataset.semantickitti.semantickitti_cylinder")
    ds = object.__new__(mod.SemkittiCylinderDataset)
    ds.training, ds.if_tta = False, False
    ds.class_names = ["c%d" % i for i in range(20)]
    ds.cylinder_space_max, ds.cylinder_space_min, ds.grid_size = np.array(fs.CYL_HI), np.array(fs.CYL_LO), np.array(fs.CYL_GRID)
    ds.point_cloud_dataset = []
    for seed in range(n_frames):
        pts = make_scan(seed, None).astype(np.float32)
        labels = np.random.default_rng(seed + 100).integers(0, 20, size=pts.shape[0]).astype(np.int64)
        ds.point_cloud_dataset.append({"labels": labels, "xyzret": pts, "path": "synthetic%d" % seed})
    b = mod.SemkittiCylinderDataset.collate_batch([ds.get_single_sample(i) for i in range(n_frames)])
    keep = {k: b[k].to(dev) for k in ("point_feature", "point_coord", "voxel_coord", "voxel_label", "point_label", "offset")}
    return (lambda: {k: v.view_as(v) for k, v in keep.items()}), int(keep["voxel_coord"].shape[0])   # new tensor objects per step


def _time_steps(step, steps, warmup):
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    if os.environ.get("PCS_BENCH_PREHEAT", "1") != "0":   # the headline's clock pre-heat (bench.py::preheat), shortened
        prev = None
        for wi in range(3):
            t0 = time.perf_counter()
            for _ in range(5):
                step()
            torch.cuda.synchronize()
            cur = time.perf_counter() - t0
            if wi >= 1 and cur >= 0.996 * prev:
                break
            prev = cur
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def bench_one(name, source, dtype, dev, steps=10, warmup=2, want_step=False):
    n_frames = int(os.environ.get("PCS_MB_FRAMES", FRAMES[name]))
    # `fuse` re-binds the glue helpers in the MODEL FILE's namespace (process-wide): put the reference's own functions back before
    # every entry, so that a `reference` entry measured after a `fuse` entry of the same file (minkunet18 after minkunet34) is unmodified
    from openpcseg_amd.block_fusion import restore_glue
    restore_glue()
    if source == "workload":
        from seeded import seeded_state
        from openpcseg_amd.workloads.minkunet import MK18_LAYERS, MK34_LAYERS, MinkUNet
        if name not in ("minkunet18", "minkunet34"):
            raise ValueError("the fused workload exists for MinkUNet only")
        model = MinkUNet(num_class=20, num_layer=MK34_LAYERS if name == "minkunet34" else MK18_LAYERS, cr=1.0)
        seeded_state(model)
    else:
        model = _reference_model(name)
    model.to(dev).train()
    if source == "fuse":
        import openpcseg_amd
        openpcseg_amd.fuse(model)
    if name == "cylinder":
        fresh, n_vox = _cylinder_batch(n_frames, dev)
    else:
        fresh, n_vox = _lidar_batch(n_frames, dev, elongation=(name == "rpvnet34"), range_view=(name == "rpvnet34"))
    params = [p for p in model.parameters() if p.requires_grad]
    opt = torch.optim.SGD(params, lr=1e-3, momentum=0.9, weight_decay=1e-4, nesterov=True)
    amp = {"bf16": torch.bfloat16, "fp16": torch.float16}.get(dtype)

    from openpcseg_amd import native

    def step():
        native.CACHE_STATS["epoch"] += 1   # memos made from here on belong to this step
        opt.zero_grad(set_to_none=True)
        if amp is None:
            ret = model(fresh())
        else:
            with torch.autocast("cuda", dtype=amp):
                ret = model(fresh())
        ret = ret[0] if isinstance(ret, tuple) else ret
        ret["loss"].backward()
        opt.step()
        return ret["loss"]
    if want_step:
        return step
    native.CACHE_STATS["cross"] = 0
    sec = _time_steps(step, steps, warmup)
    cross = native.CACHE_STATS["cross"]
    if cross and os.environ.get("PCS_MB_ALLOW_CROSS_STEP_MEMOS", "0") != "1":
        # a timed step reused a memo an earlier step left on a caller tensor (round 5: the range-image CSRs of RPVNet): not what a
        # training loop with a new batch per step pays
        raise RuntimeError("modelbench %s/%s: %d cross-step memo hits in the timed region" % (name, source, cross))
    loss = float(step().detach())
    if not np.isfinite(loss):
        raise RuntimeError("non-finite loss")
    return {"value": round(n_frames / sec, 2), "ms_per_step": round(sec * 1e3, 1), "frames": n_frames}


def run(spec, dev, steps=10, warmup=2):
    """-> {"<name>/<source>": {"f32": frames/s, "bf16": frames/s, "ms": [..], "frames": B}} (compact: one entry per model)."""
    items = AUTO if spec in ("auto", "", None) else [s for s in spec.split(",") if s]
    out = {}
    for it in items:
        parts = it.split(":")
        name = parts[0]
        source = parts[1] if len(parts) > 1 and parts[1] in ("reference", "workload", "fuse") else "reference"
        dtypes = [parts[-1]] if parts[-1] in ("f32", "bf16", "fp16") else ["f32", "bf16"]
        rec = {}
        for dt in dtypes:
            try:
                r = bench_one(name, source, dt, dev, steps, warmup)
                rec[dt] = r["value"]
                rec.setdefault("ms", []).append(r["ms_per_step"])
                rec["frames"] = r["frames"]
            except Exception as e:  # a secondary record never takes the bench line down
                rec[dt] = None
                rec["error"] = (type(e).__name__ + ": " + str(e))[:120]
            torch.cuda.empty_cache()
        out["%s/%s" % (name, {"reference": "ref", "fuse": "ref+fuse", "workload": "fused"}[source])] = rec
    return out


if __name__ == "__main__":
    print(json.dumps(run(sys.argv[1] if len(sys.argv) > 1 else "auto", torch.device("cuda:0"),
                         steps=int(os.environ.get("PCS_MB_STEPS", "10")), warmup=int(os.environ.get("PCS_MB_WARMUP", "2")))))
