"""openpcseg_amd.fuse (openpcseg_amd/block_fusion.py): block fusion applied to the reference's OWN, unmodified segmentors (R:pcseg/model/segmentor/voxel/minkunet/
minkunet.py:23-129 and its copies in spvcnn.py / rpvnet.py) must not change what they compute.

  * the fused model reproduces the reference-generated logits / loss of tests/golden/*_e2e_golden.npz (same 1e-3 bound as the
    unfused route, tests/test_reference_models.py);
  * fused vs unfused on the SAME weights: logits, loss, every parameter gradient, every BatchNorm buffer after the step;
  * state_dict keys and values unchanged; strict load both ways; unfuse() restores the classes; copy.deepcopy is independent;
  * `install_as_torchsparse(fuse=True)` fuses on the first call.

`-m "not gpu"`: through the CPU oracle backend (host logic). `-m gpu`: on libpcseg_hip.so (the fused HIP passes themselves).
Full-size: tests/test_fullsize_parity.py::test_fullsize_reference_model_fused_on_hip."""
import copy
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from stage_reference import reference_root  # noqa: E402
from test_reference_models import _Env, _load, _np  # noqa: E402

pytestmark = pytest.mark.skipif(reference_root() is None, reason="neither /root/reference nor tests/_refsrc present")


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(ROOT, "tests", "golden", "models_e2e_golden.npz"))


@pytest.fixture()
def env_oracle(monkeypatch):
    return _Env("oracle", monkeypatch)


@pytest.fixture()
def env_hip(monkeypatch, hip):
    return _Env("hip", monkeypatch)


MK34 = dict(NAME="MinkUNet", IGNORE_LABEL=0, IN_FEATURE_DIM=4, BLOCK="ResBlock", NUM_LAYER=[2, 3, 4, 6, 2, 2, 2, 2],
            PLANES=[32, 32, 64, 128, 256, 256, 128, 96, 96], cr=0.25, DROPOUT_P=0.0, LABEL_SMOOTHING=0.1, IF_DIST=False)


def _minkunet(env, **over):
    from seeded import seeded_state
    mg, mod = _load("pcseg.model.segmentor.voxel.minkunet.minkunet")
    model = mod.MinkUNet(mg._AttrDict(dict(MK34, **over)), 20)
    seeded_state(model)
    return model.to(env.dev).train()


def _mink_batch(env, g):
    from openpcseg_amd.sparse import SparseTensor
    coords = env.t(g["coords"])
    return {"lidar": SparseTensor(env.t(g["feats"]), coords), "targets": SparseTensor(env.t(g["labels"]), coords), "offset": None}


@pytest.fixture(autouse=True)
def _restore_glue():
    yield
    from openpcseg_amd.block_fusion import restore_glue
    restore_glue()


def _step(model, batch, amp=None):
    cap = {}
    # the criterion's first argument = the classifier's output in the reference's forward; a hook on the classifier itself
    # would (by design) switch the fused forward back to the reference's literal order
    h = model.criterion_losses.register_forward_pre_hook(lambda m, a: cap.__setitem__("logits", a[0].detach().float()))
    model.zero_grad(set_to_none=True)
    try:
        if amp is None:
            ret = model(batch)
        else:
            with torch.autocast("cuda", dtype=amp):
                ret = model(batch)
    finally:
        h.remove()
    ret = ret[0] if isinstance(ret, tuple) else ret
    ret["loss"].backward()
    grads = {n: p.grad.detach().double().cpu() for n, p in model.named_parameters() if p.grad is not None}
    bufs = {n: b.detach().double().cpu() for n, b in model.named_buffers()}
    return _np(cap["logits"]), float(ret["loss"].detach()), grads, bufs


def _compare_steps(a, b, logit_tol, grad_tol):
    la, lossa, ga, ba = a
    lb, lossb, gb, bb = b
    assert np.abs(la - lb).max() <= logit_tol * max(1.0, np.abs(la).max())
    assert abs(lossa - lossb) <= logit_tol * max(1.0, abs(lossa))
    assert ga.keys() == gb.keys() and ba.keys() == bb.keys()
    G = float(np.median([float(v.abs().max()) for v in ga.values()]))
    for n in ga:
        scale = max(float(ga[n].abs().max()), 1e-3 * G)   # a bias in front of a train-mode BatchNorm has a zero gradient: noise
        assert float((ga[n] - gb[n]).abs().max()) <= grad_tol * scale, n
    for n in ba:
        assert float((ba[n] - bb[n]).abs().max()) <= 1e-5 * max(1.0, float(ba[n].abs().max())), n


def _fused_vs_plain(env, g, logit_tol=2e-5, grad_tol=2e-3, amp=None, **over):
    import openpcseg_amd
    plain = _minkunet(env, **over)
    fused = _minkunet(env, **over)
    counts = openpcseg_amd.fuse(fused)
    a = _step(plain, _mink_batch(env, g), amp)
    b = _step(fused, _mink_batch(env, g), amp)
    _compare_steps(a, b, logit_tol, grad_tol)
    return counts, fused, b


# ---- CPU (oracle backend): the pass itself ----------------------------------------------------------------------------
def test_fuse_recognises_the_reference_blocks(env_oracle):
    import openpcseg_amd
    from openpcseg_amd import block_fusion as fz
    model = _minkunet(env_oracle)
    keys = list(model.state_dict().keys())
    counts = openpcseg_amd.fuse(model)
    n_res = sum(MK34["NUM_LAYER"])
    n_ds = sum(1 for m in model.modules() if hasattr(m, "downsample") and isinstance(m.downsample, torch.nn.Sequential))
    assert counts["residual"] == n_res
    assert counts["sequential"] == 1 + 4 + 4                       # stem, 4 down blocks, 4 up blocks
    assert counts["conv_bn"] == 2 + 8 + 2 * n_res + n_ds           # every Conv3d -> BatchNorm pair of the backbone
    assert counts["criterion"] == 2                                # Losses.lov_loss, Losses.ce_loss
    assert counts["glue"] == 2 and counts["forward"] == 1          # initial_voxelize + voxel_to_point; MinkUNet.forward
    import sys
    ns = sys.modules[type(model).__mro__[1].__module__]
    assert ns.voxel_to_point.__module__ == "openpcseg_amd.workloads.pointvoxel"
    assert list(model.state_dict().keys()) == keys
    assert openpcseg_amd.fuse(model) == counts                     # idempotent
    n_conv = sum(1 for m in model.modules() if type(m).__name__ == "Conv3d")
    assert sum(1 for m in model.modules() if getattr(m, "emit_bn_stats", False)) == n_conv
    assert type(model.stage1[1]).__name__ == "ResidualBlock" and type(model.stage1[1]).__mro__[1].__name__ == "ResidualBlock"
    fz.unfuse(model)
    fz.restore_glue()
    assert ns.voxel_to_point.__module__.endswith("minkunet.utils")
    assert not any(getattr(type(m), "_pcs_fused_class", False) for m in model.modules())
    assert type(model.criterion_losses.ce_loss) is torch.nn.CrossEntropyLoss
    assert list(model.state_dict().keys()) == keys


def test_forward_is_not_swapped_when_the_glue_is_not_the_references(env_oracle):
    """The replacement forwards call this package's point <-> voxel helpers: a model file whose helper was edited (or
    `fuse(glue=False)`) keeps its own forward -- and the edited helper is what runs."""
    import sys
    import openpcseg_amd
    from openpcseg_amd import block_fusion as fz
    model = _minkunet(env_oracle)
    ns = sys.modules[type(model).__module__]
    calls = []
    orig = ns.initial_voxelize

    def initial_voxelize(z, init_res, after_res):   # a fork's edit: same result, different source text
        calls.append(1)
        return orig(z, init_res, after_res)
    ns.initial_voxelize = initial_voxelize
    try:
        counts = openpcseg_amd.fuse(model)
        assert counts["forward"] == 0 and counts["glue"] == 1          # voxel_to_point alone was re-bound
        assert type(model).forward is type(model).__mro__[0].forward and not getattr(type(model), "_pcs_fused_class", False)
        assert ns.initial_voxelize is initial_voxelize
    finally:
        fz.unfuse(model)
        fz.restore_glue()
        ns.initial_voxelize = orig
    counts = openpcseg_amd.fuse(model, glue=False)
    assert counts["forward"] == 0 and counts["glue"] == 0
    fz.unfuse(model)


def test_fused_reference_minkunet_matches_the_reference_golden(golden_e2e, env_oracle):
    import openpcseg_amd
    model = _minkunet(env_oracle)
    openpcseg_amd.fuse(model)
    logits, loss, _, _ = _step(model, _mink_batch(env_oracle, golden_e2e))
    assert np.abs(logits - golden_e2e["logits"]).max() < 1e-3
    assert abs(loss - float(golden_e2e["loss"])) < 1e-3


def test_fused_equals_plain_step_on_cpu(golden_e2e, env_oracle):
    _fused_vs_plain(env_oracle, golden_e2e)


def test_fused_equals_plain_with_syncbatchnorm_classes_on_one_rank(golden_e2e, env_oracle):
    """IF_DIST=True builds the model's own SyncBatchNorm classes; without a process group they are plain batch statistics."""
    counts, fused, _ = _fused_vs_plain(env_oracle, golden_e2e, IF_DIST=True)
    assert counts["conv_bn"] > 60


def test_fused_eval_mode_equals_plain(golden_e2e, env_oracle):
    import openpcseg_amd
    from openpcseg_amd.sparse import SparseTensor
    plain, fused = _minkunet(env_oracle).eval(), _minkunet(env_oracle).eval()
    openpcseg_amd.fuse(fused)
    outs = []
    for m in (plain, fused):
        cap = {}
        h = m.classifier.register_forward_hook(lambda mod, i, o: cap.__setitem__("logits", o.detach()))
        b = _mink_batch(env_oracle, golden_e2e)
        coords = b["lidar"].C
        b.update(inverse_map=SparseTensor(torch.zeros(0, dtype=torch.long), coords[:0]), targets_mapped=SparseTensor(coords[:0, 0], coords[:0]),
                 num_points=[0], name=["x"])
        with torch.no_grad():
            try:
                m(b)
            except Exception:   # the eval branch maps predictions back per frame (host code outside the hot path)
                pass
        h.remove()
        outs.append(_np(cap["logits"]))
    assert np.abs(outs[0] - outs[1]).max() <= 2e-5 * np.abs(outs[0]).max()


def test_fused_state_dict_round_trips_and_counters(golden_e2e, env_oracle):
    import openpcseg_amd
    plain, fused = _minkunet(env_oracle), _minkunet(env_oracle)
    openpcseg_amd.fuse(fused)
    for _ in range(2):
        _step(fused, _mink_batch(env_oracle, golden_e2e))
        _step(plain, _mink_batch(env_oracle, golden_e2e))
    sd_f, sd_p = fused.state_dict(), plain.state_dict()
    assert list(sd_f.keys()) == list(sd_p.keys())
    for k in sd_p:
        if k.endswith("num_batches_tracked"):
            assert int(sd_f[k]) == int(sd_p[k]) == 2, k
    plain.load_state_dict(sd_f, strict=True)       # a checkpoint written by the fused model loads into the plain one ...
    fused.load_state_dict(plain.state_dict(), strict=True)   # ... and back


def test_deepcopy_of_a_fused_model_is_independent(golden_e2e, env_oracle):
    import openpcseg_amd
    fused = _minkunet(env_oracle)
    openpcseg_amd.fuse(fused)
    twin = copy.deepcopy(fused)
    before = {n: b.clone() for n, b in fused.named_buffers()}
    a = _step(twin, _mink_batch(env_oracle, golden_e2e))
    for n, b in fused.named_buffers():
        assert torch.equal(b, before[n]), n      # the copy ran on its own layers
    b = _step(fused, _mink_batch(env_oracle, golden_e2e))
    _compare_steps(a, b, 1e-6, 1e-5)


def test_pending_batchnorm_materialises_for_any_other_consumer(golden_e2e, env_oracle):
    """The up-convolution block's output is pending until someone reads it: `.F` gives relu(bn(conv(x)))."""
    import openpcseg_amd
    from openpcseg_amd.block_fusion import PendingBatchNorm
    from openpcseg_amd.sparse import SparseTensor, cat
    plain, fused = _minkunet(env_oracle), _minkunet(env_oracle)
    openpcseg_amd.fuse(fused)
    coords = env_oracle.t(golden_e2e["coords"])
    torch.manual_seed(3)
    feats = torch.randn(coords.shape[0], plain.stage1[0].net[0].in_channels)
    outs = []
    for m in (plain, fused):
        x = SparseTensor(feats.clone(), coords)
        x.cmaps.setdefault(x.stride, x.coords)
        d = m.stage1[0](x)                                  # down conv: builds the map the transposed conv reuses
        blk = m.up4[0]
        w = torch.randn(d.F.shape[0], blk.net[0].in_channels, generator=torch.Generator().manual_seed(4))
        up = blk(d._like(w.clone()))
        if m is fused:
            assert isinstance(up, PendingBatchNorm)
            both = cat([blk(d._like(w.clone())), x])        # fused concat
            assert both.F.shape[1] == up.F.shape[1] + x.F.shape[1]
            assert torch.allclose(both.F[:, :up.F.shape[1]], up.F, atol=1e-6) and torch.equal(both.F[:, up.F.shape[1]:], x.F)
        outs.append(up.F.detach())
    assert torch.allclose(outs[0], outs[1], atol=1e-5)


def test_auto_fuse_on_first_call(golden_e2e, env_oracle):
    from openpcseg_amd import block_fusion as fz
    model = _minkunet(env_oracle)
    fz.install_auto_fuse()
    try:
        logits, loss, _, _ = _step(model, _mink_batch(env_oracle, golden_e2e))
    finally:
        fz.uninstall_auto_fuse()
    assert model.__dict__.get("_pcs_fused") is not None and model.__dict__["_pcs_fused"]["counts"]["residual"] == sum(MK34["NUM_LAYER"])
    assert np.abs(logits - golden_e2e["logits"]).max() < 1e-3


def test_fused_spvcnn_and_rpvnet_match_their_goldens(gold, env_oracle):
    """The block copies in fusion/spvcnn/spvcnn.py and fusion/rpvnet/rpvnet.py are recognised too."""
    import test_reference_models as trm
    from openpcseg_amd import block_fusion as fz
    fz.install_auto_fuse()
    try:
        trm._run_spvcnn(env_oracle, gold)
        trm._run_rpvnet(env_oracle, gold)
        trm._run_cylinder(env_oracle, gold)     # no recognised block: must simply keep working
    finally:
        fz.uninstall_auto_fuse()


# ---- HIP --------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_fused_reference_minkunet_on_hip(golden_e2e, env_hip):
    import openpcseg_amd
    model = _minkunet(env_hip)
    openpcseg_amd.fuse(model)
    logits, loss, _, _ = _step(model, _mink_batch(env_hip, golden_e2e))
    assert np.abs(logits - golden_e2e["logits"]).max() < 1e-3
    assert abs(loss - float(golden_e2e["loss"])) < 1e-3


def _eval_logits(env, g, fuse):
    import openpcseg_amd
    from openpcseg_amd.sparse import SparseTensor
    m = _minkunet(env).eval()
    if fuse:
        openpcseg_amd.fuse(m)
    cap = {}
    h = m.classifier.register_forward_hook(lambda mod, i, o: cap.__setitem__("logits", o.detach().float()))
    b = _mink_batch(env, g)
    coords = b["lidar"].C
    b.update(inverse_map=SparseTensor(torch.zeros(0, dtype=torch.long, device=coords.device), coords[:0]),
             targets_mapped=SparseTensor(coords[:0, 0], coords[:0]), num_points=[0], name=["x"])
    with torch.no_grad():
        try:
            m(b)
        except Exception:   # the eval branch maps predictions back per frame on the host (outside the hot path)
            pass
    h.remove()
    return _np(cap["logits"])


@pytest.mark.gpu
def test_fused_eval_mode_equals_plain_on_hip(golden_e2e, env_hip):
    """Inference (`model.eval()`, `torch.no_grad()`): the fused blocks normalise with the running statistics like nn.BatchNorm1d."""
    a, b = _eval_logits(env_hip, golden_e2e, False), _eval_logits(env_hip, golden_e2e, True)
    assert np.abs(a - b).max() <= 5e-5 * np.abs(a).max()


@pytest.mark.gpu
def test_fused_equals_plain_step_on_hip(golden_e2e, env_hip):
    _fused_vs_plain(env_hip, golden_e2e, logit_tol=5e-5, grad_tol=2e-2)   # BatchNorm bias gradients: signed sums that nearly cancel


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
def test_fused_equals_plain_step_on_hip_bf16(golden_e2e, env_hip, dtype):
    """Under autocast (the reference's `--amp`: fp16; bf16 as well) both routes store 16-bit activations; the fused passes round once
    where torch rounds per op. Compared with each other and with the reference's fp32 logits."""
    import openpcseg_amd
    plain, fused = _minkunet(env_hip), _minkunet(env_hip)
    openpcseg_amd.fuse(fused)
    a = _step(plain, _mink_batch(env_hip, golden_e2e), dtype)
    b = _step(fused, _mink_batch(env_hip, golden_e2e), dtype)
    rms = float(np.sqrt((a[0].astype(np.float64) ** 2).mean()))
    k = 1.0 if dtype == torch.bfloat16 else 0.25          # fp16 keeps three more mantissa bits
    assert np.abs(a[0] - b[0]).max() < 0.35 * k * rms and np.abs(a[0] - b[0]).mean() < 0.03 * k * rms
    assert (a[0].argmax(1) == b[0].argmax(1)).mean() > 0.95
    assert abs(a[1] - b[1]) < 0.03 * abs(a[1])
    ref = golden_e2e["logits"]
    assert np.abs(b[0] - ref).mean() < 0.03 * k * rms and np.isfinite(b[1])
    assert all(torch.isfinite(g).all() for g in b[2].values())


@pytest.mark.gpu
def test_fused_spvcnn_and_rpvnet_on_hip(gold, env_hip):
    import test_reference_models as trm
    from openpcseg_amd import block_fusion as fz
    fz.install_auto_fuse()
    try:
        trm._run_spvcnn(env_hip, gold)
        trm._run_rpvnet(env_hip, gold)
        trm._run_rpvnet(env_hip, gold, **trm.WAYMO)
        trm._run_cylinder(env_hip, gold)
    finally:
        fz.uninstall_auto_fuse()


@pytest.mark.gpu
@pytest.mark.parametrize("amp", [None, torch.bfloat16])
def test_fused_cylinder_blocks_equal_plain_on_hip(env_hip, monkeypatch, amp):
    """Cylinder_TS's ResContextBlock / ResBlock / UpBlock (R:pcseg/model/segmentor/voxel/cylinder3d/cylinder_ts.py:88-330) after
    `fuse`: LeakyReLU and the BatchNorm statistics in the convolution's write-back, the activation's derivative in the BatchNorm
    backward apply pass -- against the same blocks with that fusion switched off (PCS_CYL_FUSED=0: the reference's own op sequence on
    this package's kernels). Outputs, input gradient, every parameter gradient and the BatchNorm buffers."""
    import openpcseg_amd
    from openpcseg_amd import block_fusion as fz
    from openpcseg_amd.sparse import SparseTensor
    from openpcseg_amd.workloads.synthetic import make_batch
    mg, mod = _load("pcseg.model.segmentor.voxel.cylinder3d.cylinder_ts")
    coords = make_batch([3], n_points=20000)["lidar"].C.to(env_hip.dev)
    n = coords.shape[0]

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.ctx = mod.ResContextBlock(32, 32, indice_key="pre")
            self.down = mod.ResBlock(32, 64, 0.2, height_pooling=True, indice_key="down2")
            self.up = mod.UpBlock(64, 64, indice_key="up0", up_key="down2", height_pooling=True)

        def forward(self, x):
            c = self.ctx(x)
            d, skip = self.down(c)
            return self.up(d, skip)

    def run(fused):
        monkeypatch.setenv("PCS_CYL_FUSED", "1" if fused else "0")
        torch.manual_seed(11)
        net = Net().to(env_hip.dev).train()
        counts = openpcseg_amd.fuse(net)
        assert counts["cylinder"] == 3
        x = torch.randn(n, 32, device=env_hip.dev, generator=torch.Generator(device=env_hip.dev).manual_seed(2)).requires_grad_(True)
        with torch.autocast("cuda", dtype=amp, enabled=amp is not None):
            out = net(SparseTensor(x, coords))
        gy = torch.randn(out.F.shape, device=env_hip.dev, generator=torch.Generator(device=env_hip.dev).manual_seed(4))
        out.F.float().backward(gy)
        res = {"out": out.F.float().detach(), "gx": x.grad}
        res.update({"g:" + k: p.grad for k, p in net.named_parameters() if p.grad is not None})
        res.update({"b:" + k: b.float() for k, b in net.named_buffers() if b.dtype.is_floating_point})
        fz.unfuse(net)
        return res

    plain, fused = run(False), run(True)
    assert plain.keys() == fused.keys()

    def err(a, b):
        return {k: float((a[k].float() - b[k].float()).abs().mean()) / max(float(b[k].float().abs().mean()), 1e-9) for k in b}
    if amp is None:
        worst = {k: float((fused[k] - plain[k]).abs().max()) / max(float(plain[k].abs().max()), 1e-6) for k in plain}
        assert max(worst.values()) <= 2e-4, max(worst.items(), key=lambda kv: kv[1])
    else:
        # two 16-bit routes that round at different places drift apart through a dozen BatchNorm backward passes (cancellation in
        # g - mean(g) - xhat mean(g xhat)): each is held against the SAME blocks in fp32 -- the fused route must be as close to
        # fp32 as the plain 16-bit route is
        amp = None
        ref = run(False)
        e_plain, e_fused = err(plain, ref), err(fused, ref)
        if os.environ.get("PCS_TEST_VERBOSE"):
            for k in ref:
                print("%-40s plain %.2e fused %.2e" % (k, e_plain[k], e_fused[k]))
        for k in ref:
            assert e_fused[k] <= 1.5 * e_plain[k] + 2e-3, (k, e_fused[k], e_plain[k])


@pytest.mark.gpu
@pytest.mark.parametrize("cin,cout", [(9, 64), (32, 256), (256, 20), (100, 36)])
def test_dense_linear_and_batchnorm_match_torch_on_hip(hip, cin, cout):
    """The re-classed stock nn.Linear / nn.BatchNorm1d of a fused model ((N, C) rows of >= 4096 points: Cylinder_TS's point MLP and
    per-convolution BatchNorm1d, the point_transforms of SPVCNN / RPVNet) against the stock modules: outputs, input / weight / bias
    gradients, running statistics; and that the fused path is really taken."""
    import openpcseg_amd
    from openpcseg_amd import block_fusion as fz
    from openpcseg_amd import modules as spnn

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.marker = spnn.Conv3d(4, 4, 1)        # a sparse model: fuse() only touches dense layers of models that have sparse convolutions
            self.bn_in = torch.nn.BatchNorm1d(cin)
            self.mlp = torch.nn.Sequential(torch.nn.Linear(cin, cout), torch.nn.BatchNorm1d(cout), torch.nn.ReLU(True))
            self.head = torch.nn.Linear(cout, 20)
            self.stem = torch.nn.Sequential(spnn.Conv3d(4, 8, 3), spnn.BatchNorm(8), spnn.ReLU(True))

        def forward(self, x):
            return self.head(self.mlp(self.bn_in(x)))

    torch.manual_seed(3)
    plain, fused = Net().cuda().train(), Net().cuda().train()
    fused.load_state_dict(plain.state_dict())
    counts = openpcseg_amd.fuse(fused)
    assert counts["dense"] == 2 and counts["sequential"] == 2 and counts["conv_bn"] == 2, counts
    x = torch.randn(30000, cin, device="cuda") * 2 + 0.5
    calls = []
    orig = fz._PointLinear.apply
    fz._PointLinear.apply = staticmethod(lambda *a: (calls.append(1), orig(*a))[1])
    try:
        outs = []
        for m in (plain, fused):
            xi = x.clone().requires_grad_(True)
            y = m(xi)
            (y * torch.arange(1, 21, device="cuda")).sum().backward()
            outs.append((y.detach(), xi.grad, {n: p.grad for n, p in m.named_parameters() if p.grad is not None},
                         {n: b.clone() for n, b in m.named_buffers()}))
    finally:
        fz._PointLinear.apply = orig
    assert len(calls) == 2
    (y0, g0, p0, b0), (y1, g1, p1, b1) = outs
    assert float((y0 - y1).abs().max()) <= 2e-5 * float(y0.abs().max())
    assert float((g0 - g1).abs().max()) <= 1e-4 * float(g0.abs().max())
    for n in p0:
        if n in ("bn_in.bias", "mlp.0.bias"):   # a constant shift in front of a train-mode BatchNorm: gradient zero by construction (noise)
            continue
        assert float((p0[n] - p1[n]).abs().max()) <= 2e-4 * max(float(p0[n].abs().max()), 1e-3), n
    for n in b0:
        assert torch.allclose(b0[n].float(), b1[n].float(), rtol=1e-5, atol=1e-6), n
    # small inputs and eval mode take the stock forward / the running statistics
    small = torch.randn(100, cin, device="cuda")
    assert torch.allclose(plain.eval()(small), fused.eval()(small), atol=1e-5)
    assert torch.allclose(plain(x), fused(x), atol=2e-5 * float(y0.abs().max()))


def test_checkpoint_round_trip_in_the_reference_format(golden_e2e, env_oracle, tmp_path):
    """R:train.py:285-301 / :303-318 (`save_checkpoint` / `resume`): a checkpoint written from the FUSED model -- model_state through
    the reference's own `model_state_to_cpu`, optimizer / scaler / scheduler states -- is resumed by the PLAIN reference model through
    the reference's own `load_params(..., strict=True)`, and the next training step of both gives the same loss and the same weights;
    this package's MinkUNet workload loads the same file strictly as well."""
    import openpcseg_amd
    from openpcseg_amd.workloads.minkunet import MinkUNet as WorkloadMinkUNet
    try:
        from tools.utils.train_utils import model_state_to_cpu
    except Exception:   # the staged copy on the GPU box may not carry tools/utils/train_utils.py (R:tools/utils/train_utils.py:138-142)
        def model_state_to_cpu(ms):
            return type(ms)((k, v.cpu()) for k, v in ms.items())

    def rig(model):
        opt = torch.optim.SGD(model.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
        return opt, torch.optim.lr_scheduler.LambdaLR(opt, lambda it: 1.0 / (1 + it)), torch.amp.GradScaler("cuda", enabled=False)

    def step(model, opt, sched, scaler):
        model.train()
        opt.zero_grad()
        ret = model(_mink_batch(env_oracle, golden_e2e))
        loss = ret[0]["loss"].mean()
        scaler.scale(loss).backward()
        scaler.unscale_(opt)
        torch.nn.utils.clip_grad_norm_(model.parameters(), 10.0)
        scaler.step(opt)
        scaler.update()
        sched.step()
        return float(loss.detach())

    fused = _minkunet(env_oracle)
    openpcseg_amd.fuse(fused)
    opt, sched, scaler = rig(fused)
    for _ in range(2):
        step(fused, opt, sched, scaler)
    ckpt = {"epoch": 1, "it": 2, "model_state": model_state_to_cpu(fused.state_dict()), "optimizer_state": opt.state_dict(),
            "scaler_state": scaler.state_dict(), "scheduler_state": sched.state_dict()}
    path = str(tmp_path / "checkpoint_epoch_1.pth")
    torch.save(ckpt, path)
    disk = torch.load(path, map_location="cpu")

    plain = _minkunet(env_oracle)
    for p in plain.parameters():
        p.data.add_(1.0)                                   # nothing may survive from the construction
    msg = plain.load_params(disk["model_state"], strict=True)   # R:pcseg/model/segmentor/base_segmentors.py:16-26
    assert not msg.missing_keys and not msg.unexpected_keys
    opt2, sched2, scaler2 = rig(plain)
    opt2.load_state_dict(disk["optimizer_state"])
    scaler2.load_state_dict(disk["scaler_state"])
    sched2.load_state_dict(disk["scheduler_state"])
    la, lb = step(fused, opt, sched, scaler), step(plain, opt2, sched2, scaler2)
    assert abs(la - lb) <= 2e-5 * abs(la)
    sa, sb = fused.state_dict(), plain.state_dict()
    assert list(sa.keys()) == list(sb.keys())
    for k in sa:
        assert torch.allclose(sa[k].double(), sb[k].double(), rtol=2e-4, atol=2e-5), k
    wl = WorkloadMinkUNet(num_class=20, num_layer=MK34["NUM_LAYER"], cr=MK34["cr"])
    wl.load_state_dict(disk["model_state"], strict=True)


def test_bottleneck_block_is_recognised_too(golden_e2e, env_oracle):
    """R:pcseg/model/segmentor/voxel/minkunet/minkunet.py:132-186 (the `Bottleneck` block, the default of `BLOCK` when a config does not
    name one): 1x1x1 -> k3 -> 1x1x1 convolutions with BatchNorms and no inner ReLU, `relu(net(x) + downsample(x))`. Fused vs plain on
    the same weights: output, input gradient, every parameter gradient, BatchNorm buffers."""
    import openpcseg_amd
    from openpcseg_amd.sparse import SparseTensor
    from seeded import seeded_state
    _, mod = _load("pcseg.model.segmentor.voxel.minkunet.minkunet")
    coords = env_oracle.t(golden_e2e["coords"])
    torch.manual_seed(2)
    feats = torch.randn(coords.shape[0], 16)
    outs = []
    for fuse in (False, True):
        blk = mod.Bottleneck(16, 8)          # 16 -> 8 x 4 = 32 channels: the variant with a downsample branch
        seeded_state(blk)
        blk.train()
        if fuse:
            counts = openpcseg_amd.fuse(blk)
            assert counts["residual"] == 1 and counts["conv_bn"] == 4, counts
        x = feats.clone().requires_grad_(True)
        y = blk(SparseTensor(x, coords)).F
        (y * torch.arange(1, 33)).sum().backward()
        outs.append((y.detach(), x.grad, {n: p.grad for n, p in blk.named_parameters() if p.grad is not None},
                     {n: b.clone() for n, b in blk.named_buffers()}))
    (y0, g0, p0, b0), (y1, g1, p1, b1) = outs
    assert torch.allclose(y0, y1, atol=2e-5 * float(y0.abs().max()))
    assert torch.allclose(g0, g1, atol=1e-4 * float(g0.abs().max()))
    G = float(np.median([float(v.abs().max()) for v in p0.values()]))
    for n in p0:
        assert float((p0[n] - p1[n]).abs().max()) <= 2e-3 * max(float(p0[n].abs().max()), 1e-3 * G), n
    for n in b0:
        assert torch.allclose(b0[n].float(), b1[n].float(), rtol=1e-5, atol=1e-6), n
