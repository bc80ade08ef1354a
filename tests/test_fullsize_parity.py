"""BASELINE configs 2-5 at full size, forward AND backward, against the reference run on its own CPU backend.

Fixtures: tests/golden/config{2,3,4,5}_golden.npz, made by `make_golden.py configN` by RUNNING the reference's segmentor
(train mode, one full 120 000-ray synthetic frame, fp32, forward + loss + backward; tests/golden/fullsize.py holds the
definitions shared with the generator). Each `-m gpu` test rebuilds the same frame from its seed (CRC-checked), runs
the same model with the same seeded weights through libpcseg_hip.so and compares

  * per-point logits (every 16th row element-wise; float64 column sums over all rows), loss;
  * per-parameter gradients after loss.backward() (R:train.py:360-371): float64 abs-sum / sum / eight samples each;

for (a) the reference's own model source on the HIP backend (configs 2-5) and (b) this package's fused MinkUNet workload
-- the graph bench.py times (BN statistics from the conv epilogue, strided-dy concat backward, column-block classifier)
-- name-mapped onto config 2's reference gradients, in fp32 and under bf16 autocast.

Bounds. north_star asks "per-point logits within 1e-3 fp32": measured 1.4e-4 ... 8.8e-4 ABSOLUTE at |logit| up to
109 / 94 / 19 / 281 (configs 2 / 3 / 4 / 5), i.e. ~1.5e-6 of the logit scale after 40-120 layers of MFMA-vs-scalar
summation order. Every bound below is absolute for logits and loss, sits next to the value measured on MI355X
(profiles/round3_fullsize_parity.json, written by this test) and is at most ~2x that value.
"""
import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import fullsize as fs  # noqa: E402
from stage_reference import reference_root  # noqa: E402

pytestmark = pytest.mark.skipif(reference_root() is None, reason="neither /root/reference nor tests/_refsrc present")

# Bounds (absolute for logits / loss, relative for gradients), each at most ~2x the value measured on MI355X in round 3
# (profiles/round3_fullsize_parity.json holds the measured values). Gradients: "matrix" = every >= 2-D parameter
# (convolution / linear kernels: sums over ~1e5 rows without cancellation); "any" = all live parameters, including
# BatchNorm scales / biases and layer biases whose gradients are sums of signed terms that nearly cancel (the fp32
# summation order of EITHER side shows at the 1e-3 level there; stem.1.bias is the worst).
# Config 4 (Cylinder_TS) is looser and says why: (1) its first layer is torch's own BatchNorm1d over RAW point features
# (rho ~ 20 m, z ~ -2 m: E[x^2] - mean^2 in fp32) -- torch's GPU kernel and torch's CPU kernel differ by 2.9e-5 relative
# right there, before any of this package's code runs (tools/module_trace.py, profiles/round3_config4_module_trace.txt);
# (2) the fixture's torch_scatter.scatter_max is a scatter_reduce('amax') stand-in (torch_scatter is not installed in the
# build container: parity unpinned for that op), whose backward splits a tie evenly where scatter_max picks one index.
BOUNDS = {
    # name: (logit max-abs err, loss abs err, matrix abs-sum rel, matrix sample / abs-max, any abs-sum rel, any sample / abs-max)
    #   measured:      logits   matrix abs-sum / sample    any abs-sum / sample
    "config2/reference": (3e-4, 1e-5, 6e-4, 2.5e-3, 6e-3, 1.1e-2),   # 1.5e-4   2.8e-4 / 1.2e-3   2.9e-3 / 5.4e-3
    "config2/workload": (3e-4, 1e-5, 1e-4, 6e-4, 1e-3, 1e-2),        # 1.4e-4   4.1e-5 / 2.8e-4   4.8e-4 / 4.7e-3
    "config3/reference": (4e-4, 1e-5, 3e-4, 7e-4, 5e-4, 3e-3),       # 1.7e-4   1.4e-4 / 3.5e-4   2.1e-4 / 1.4e-3
    "config4/reference": (1e-3, 1e-5, 3e-3, 3.6e-2, 1.8e-2, 5.5e-2),  # 8.8e-4   1.5e-3 / 1.8e-2   8.8e-3 / 2.7e-2 (see above)
    "config2x2/reference": (9e-4, 1e-5, 9e-4, 4e-3, 4e-3, 1.3e-2),    # 4.5e-4   4.5e-4 / 2.0e-3   1.9e-3 / 6.7e-3 (two-frame batch, logit scale 150; r4)
    "config5/reference": (8e-4, 2e-5, 2e-4, 6e-4, 5e-4, 4e-3),       # 4.1e-4   7.8e-5 / 2.9e-4   2.3e-4 / 1.8e-3
    # [r5] the graph the headline metric is quoted on: MinkUNet-34 cr1.0 (NUM_LAYER [2,3,4,6,2,2,2,2]), one full frame (seed 6),
    # logit scale 318 (3x config 2's)
    "config_mk34/reference": (8e-4, 1e-5, 2e-4, 3.6e-3, 2e-3, 5e-3),      # 3.9e-4   7.3e-5 / 1.8e-3   9.0e-4 / 2.6e-3 (+fuse: 4.1e-4, 9.0e-5 / 1.8e-3, 8.7e-4 / 2.5e-3)
    "config_mk34/workload": (8e-4, 1e-5, 2e-4, 3.6e-3, 2e-3, 5e-3),
}
# the same reference sources after openpcseg_amd.fuse(model) (block fusion, openpcseg_amd/block_fusion.py): the bounds of the plain route
for _k in ("config2", "config3", "config4", "config2x2", "config_mk34"):
    BOUNDS[_k + "/reference+fuse"] = BOUNDS[_k + "/reference"]
# config 5 after fuse: the point MLPs' BatchNorm runs on the fused passes (another summation order than torch's kernels), measured
#                                                                  4.5e-4   1.9e-4 / 3.2e-4   5.1e-4 / 1.7e-3
BOUNDS["config5/reference+fuse"] = (8e-4, 2e-5, 4e-4, 6e-4, 1e-3, 4e-3)
_MEASURED = {}


def _record(name, m):
    _MEASURED[name] = m
    print("\n[fullsize parity] %s: %s" % (name, json.dumps(m)))
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "fullsize_parity_measured.json"), "w") as f:
            json.dump(_MEASURED, f, indent=1, sort_keys=True)


def _assert_bounds(name, m):
    lo, ls, gm, gms, ga, gas = BOUNDS[name]
    assert m["logit_max_abs_err"] < lo, (name, m)
    assert m["colsum_err_per_row"] < lo, (name, m)
    assert m["abssum_rel_err"] < 1e-4, (name, m)
    assert m["loss_abs_err"] < ls * max(1.0, abs(m.get("loss_ref", 1.0))), (name, m)   # measured <= 7.6e-6 at |loss| <= 37
    assert m["grad_matrix_abssum_rel_err"] < gm and m["grad_matrix_sample_err_rel_max"] < gms, (name, m)
    assert m["grad_abssum_rel_err"] < ga and m["grad_sample_err_rel_max"] < gas, (name, m)
    assert m["grad_dead_max_over_G"] < 1e-3, (name, m)   # gradients that are zero by construction stay rounding noise


def _golden(cfg):
    p = os.path.join(ROOT, "tests", "golden", "%s_golden.npz" % cfg)
    if not os.path.exists(p):
        pytest.skip("%s not generated" % os.path.basename(p))
    g = np.load(p)
    if int(g["n_points"]) != -1:
        pytest.skip("%s holds a reduced frame (%d rays)" % (os.path.basename(p), int(g["n_points"])))
    return g


def _inputs(cfg, g):
    """The frame(s) the reference saw, rebuilt from the seed(s); CRCs of every input array must match the fixture."""
    import openpcseg_amd
    from openpcseg_amd.sparse import SparseTensor
    openpcseg_amd.install_reference_aliases()
    if cfg == "config4":
        import make_golden
        make_golden.import_reference_minkunet()  # import stubs + sys.path for the (staged) reference tree
    batch = fs.build_inputs(cfg, SparseTensor)
    for k, v in fs.input_crcs(cfg, batch).items():
        assert int(v) == int(g[k]), "input %s differs from the frame the reference ran on" % k
    return batch


def _reference_model(cfg):
    import make_golden as mg
    import openpcseg_amd
    from seeded import seeded_state
    openpcseg_amd.install_reference_aliases()
    dotted, cls = fs.MODEL_PATH[cfg]
    mod = mg.import_reference_model(dotted)
    for m in list(sys.modules.values()):  # names bound to import placeholders by an earlier import in this process
        if getattr(m, "__name__", "").startswith(("pcseg.", "tools.")) and hasattr(m, "torch_scatter"):
            m.torch_scatter = sys.modules["torch_scatter"]
    if cfg == "config5":
        mod.rnf = sys.modules["range_utils.nn.functional"]
    model = getattr(mod, cls)(mg._AttrDict(fs.MODEL_CFG[cfg]), 20)
    seeded_state(model)
    return model


@pytest.mark.parametrize("cfg", ["config2", "config3", "config4", "config5", "config2x2", "config_mk34"])
def test_fullsize_inputs_regenerate(cfg):
    """CPU: the seeded frame of every fixture regenerates bit-identically (what makes the fixtures usable at all)."""
    _inputs(cfg, _golden(cfg))


def _reference_step_on_hip(cfg, fuse):
    import openpcseg_amd
    from openpcseg_amd.sparse import SparseTensor
    g = _golden(cfg)
    dev = torch.device("cuda:0")
    batch = fs.to_device(cfg, _inputs(cfg, g), dev, SparseTensor)
    model = fs.freeze_dropout(_reference_model(cfg).to(dev).train())
    if fuse:
        counts = openpcseg_amd.fuse(model)
        if cfg == "config4":   # Cylinder_TS: no Sequential / residual blocks of the MinkUNet kind -- its BatchNorm1d / Linear / loss modules
            assert counts["dense"] >= 40 and counts["criterion"] == 3, counts
        else:
            assert counts["residual"] >= 16 and counts["criterion"] == 2 and counts["glue"] >= 2, counts
        assert counts["forward"] == (1 if fs.MODEL_PATH[cfg][1] in ("MinkUNet", "SPVCNN") else 0), counts
    try:
        logits, loss = fs.run_train_step(cfg, model, batch, via="criterion" if fuse else "classifier")
    finally:
        if fuse:
            from openpcseg_amd.block_fusion import restore_glue
            restore_glue()
    m = fs.compare(g, logits, loss, fs.model_grads(model))
    m["loss_ref"] = float(g["loss"])
    name = cfg + ("/reference+fuse" if fuse else "/reference")
    _record(name, m)
    _assert_bounds(name, m)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", ["config2", "config3", "config4", "config5", "config2x2", "config_mk34"])
def test_fullsize_reference_model_on_hip(cfg, hip):
    """The reference's own segmentor source on libpcseg_hip.so: logits, loss and every parameter gradient of one full
    training step vs the reference's run on its own CPU backend."""
    _reference_step_on_hip(cfg, fuse=False)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", ["config2", "config3", "config4", "config5", "config2x2", "config_mk34"])
def test_fullsize_reference_model_fused_on_hip(cfg, hip):
    """The same sources after `openpcseg_amd.fuse(model)` -- conv-epilogue BatchNorm statistics, BN + residual + ReLU in one pass,
    concat written by the apply pass, device Lovasz-softmax / masked-mean CE -- against the same fixtures with the same bounds."""
    _reference_step_on_hip(cfg, fuse=True)


def _workload(g, dev, amp=None, wgrad="fp32", conv="fp32", cfg="config2"):
    from seeded import seeded_state
    from openpcseg_amd import functional as pcsF
    from openpcseg_amd.sparse import SparseTensor
    from openpcseg_amd.workloads.minkunet import MinkUNet
    batch = fs.to_device(cfg, _inputs(cfg, g), dev, SparseTensor)
    model = MinkUNet(num_class=20, num_layer=fs.MODEL_CFG[cfg]["NUM_LAYER"], cr=1.0)
    seeded_state(model)
    model.to(dev).train()
    pcsF.set_wgrad_policy(wgrad)
    pcsF.set_conv_policy(conv)
    try:
        if amp is None:
            out = model(batch)
        else:
            with torch.autocast("cuda", dtype=amp):
                out = model(batch)
        out["loss"].backward()
    finally:
        pcsF.set_wgrad_policy("fp32")
        pcsF.set_conv_policy("fp32")
    return out["logits"].detach().float().cpu().numpy(), float(out["loss"].detach()), fs.model_grads(model)


@pytest.mark.gpu
@pytest.mark.parametrize("wgrad", ["fp32", "bf16x3", "bf16x3+conv"])
def test_fullsize_workload_minkunet18_on_hip(hip, wgrad):
    """Config 2 through this package's fused MinkUNet workload (what bench.py times): same frame, same weights, same
    reference logits AND gradients -- the autograd wiring of the fused graph (conv-epilogue BN statistics, strided-dy
    concat backward, column-block classifier) against the reference's plain graph. wgrad = "bf16x3" is the policy of
    bench.py's fp32 line (three-plane split weight gradient on the >= 96-channel layers); "bf16x3+conv" adds the split
    forward / input-gradient kernel (bench.py's fp32_bf16x3 record): the same bounds hold for all three."""
    g = _golden("config2")
    logits, loss, grads = _workload(g, torch.device("cuda:0"), wgrad=wgrad.split("+")[0],
                                    conv="bf16x3" if wgrad.endswith("+conv") else "fp32")
    m = fs.compare(g, logits, loss, grads)
    m["loss_ref"] = float(g["loss"])
    _record("config2/workload" + ("" if wgrad == "fp32" else "/wgrad-" + wgrad), m)
    _assert_bounds("config2/workload", m)


@pytest.mark.gpu
def test_fullsize_workload_minkunet34_on_hip(hip):
    """[r5] The exact graph bench.py's headline times -- this package's fused MinkUNet with MK34_LAYERS, cr 1.0 -- against the
    reference's MinkUNet-34 run (tests/golden/config_mk34_golden.npz: R:tools/cfgs/voxel/semantic_kitti/minkunet_mk34_cr10.yaml)."""
    g = _golden("config_mk34")
    logits, loss, grads = _workload(g, torch.device("cuda:0"), cfg="config_mk34")
    m = fs.compare(g, logits, loss, grads)
    m["loss_ref"] = float(g["loss"])
    _record("config_mk34/workload", m)
    _assert_bounds("config_mk34/workload", m)


# bf16 / fp16 autocast: 16-bit storage of every activation (8 / 11 significant bits), fp32 accumulation. The bound is per
# point, relative to the RMS of the reference logits (measured: see profiles/round3_fullsize_parity.json).
# measured (bf16 / fp16): max 0.21 / 0.026 of the RMS, mean 0.0098 / 0.0013, worst parameter-gradient abs-sum 8.7 % / 2.8 %,
# arg-max agreement with the fp32 reference 98.6 % / 99.7 % of the points
AMP_BOUNDS = {torch.bfloat16: (0.30, 0.02, 0.12, 0.975), torch.float16: (0.05, 0.0025, 0.04, 0.99)}  # max / rms, mean / rms, grad, arg-max
# [r5] MinkUNet-34 (the headline graph, 15 more residual blocks than config 2)
#   measured [r5] (bf16 / fp16): max 0.288 / 0.038 of the RMS, mean 0.0060 / 0.00073, worst gradient abs-sum 20.7 % / 7.5 %, arg-max 98.3 / 99.9 %
AMP_BOUNDS_MK34 = {torch.bfloat16: (0.45, 0.012, 0.30, 0.97), torch.float16: (0.06, 0.0015, 0.14, 0.99)}


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", ["config2", "config_mk34"])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
def test_fullsize_workload_autocast_vs_fp32_reference(dtype, cfg, hip):
    """Model-level half-precision check against the fp32 REFERENCE logits and gradients (not against our own fp32 run):
    a wrong layer, a dropped residual or a mis-scaled gradient in the 16-bit path moves these by O(1)."""
    g = _golden(cfg)
    logits, loss, grads = _workload(g, torch.device("cuda:0"), amp=dtype, cfg=cfg)
    step, ref = int(g["row_step"]), g["logits_rows"]
    rms = float(np.sqrt((ref.astype(np.float64) ** 2).mean()))
    err = np.abs(logits[::step] - ref)
    m = fs.compare(g, logits, loss, grads)
    m.update({"logit_rms": rms, "logit_max_err_over_rms": float(err.max() / rms), "logit_mean_err_over_rms": float(err.mean() / rms),
              "argmax_agreement": float((logits[::step].argmax(1) == ref.argmax(1)).mean()), "loss_ref": float(g["loss"])})
    _record(cfg + "/workload/" + str(dtype).split(".")[1], m)
    bmax, bmean, bgrad, bagree = (AMP_BOUNDS if cfg == "config2" else AMP_BOUNDS_MK34)[dtype]
    assert m["logit_max_err_over_rms"] < bmax, m
    assert m["logit_mean_err_over_rms"] < bmean, m
    assert m["argmax_agreement"] > bagree, m
    assert m["grad_abssum_rel_err"] < bgrad, m
    assert m["loss_abs_err"] < bmean * abs(float(g["loss"])), m


# The reference trains under --amp (R:dist_train.sh:18). Its SPVCNN (config 3) and RPVNet mk34 cr 1.75 (config 5) sources under
# torch.autocast on the HIP backend against THEIR fp32 fixtures: the convolutions run on the 16-bit MFMA kernels (incl. the TAIL
# instances of the cr 1.75 widths), BatchNorm / point ops / the range branch follow torch's autocast rules. Bounds per point
# relative to the RMS of the reference logits, as AMP_BOUNDS above but for a graph with fp16-unfriendly pieces this package does
# not own (point MLPs, the SalsaNext range branch): max / rms, mean / rms, worst live parameter-gradient abs-sum, arg-max agreement.
# (max err / rms, mean err / rms, gradient abs-sum rel, arg-max agreement), each <= 2x the value measured on MI355X in round 4:
#   config3  0.187 / 0.0092 / 0.072 / 0.988      config5  0.158 / 0.0076 / 0.101 / 0.991  (profiles/round4_fullsize_parity.json)
REF_AMP_BOUNDS = {"config3": (0.38, 0.019, 0.15, 0.975), "config5": (0.32, 0.016, 0.20, 0.982)}


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", ["config3", "config5"])
def test_fullsize_reference_model_autocast_bf16(cfg, hip):
    from openpcseg_amd.sparse import SparseTensor
    g = _golden(cfg)
    dev = torch.device("cuda:0")
    batch = fs.to_device(cfg, _inputs(cfg, g), dev, SparseTensor)
    model = fs.freeze_dropout(_reference_model(cfg).to(dev).train())
    with torch.autocast("cuda", dtype=torch.bfloat16):
        logits, loss = fs.run_train_step(cfg, model, batch)
    logits = logits.astype(np.float32)
    step, ref = int(g["row_step"]), g["logits_rows"]
    rms = float(np.sqrt((ref.astype(np.float64) ** 2).mean()))
    err = np.abs(logits[::step] - ref)
    m = fs.compare(g, logits, loss, fs.model_grads(model))
    m.update({"logit_rms": rms, "logit_max_err_over_rms": float(err.max() / rms), "logit_mean_err_over_rms": float(err.mean() / rms),
              "argmax_agreement": float((logits[::step].argmax(1) == ref.argmax(1)).mean()), "loss_ref": float(g["loss"])})
    _record(cfg + "/reference/bf16", m)
    bmax, bmean, bgrad, bagree = REF_AMP_BOUNDS[cfg]
    assert np.isfinite(logits).all() and np.isfinite(loss)
    assert m["logit_max_err_over_rms"] < bmax, m
    assert m["logit_mean_err_over_rms"] < bmean, m
    assert m["argmax_agreement"] > bagree, m
    assert m["grad_abssum_rel_err"] < bgrad, m
