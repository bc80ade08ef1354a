"""Block fusion for the reference's OWN, unmodified segmentors (SURVEY.md section 8f-2 on the route north_star names).

The reference builds its backbones from three blocks (R:pcseg/model/segmentor/voxel/minkunet/minkunet.py:23-129, identical
copies in fusion/spvcnn/spvcnn.py and fusion/rpvnet/rpvnet.py):

    BasicConvolutionBlock / BasicDeconvolutionBlock   nn.Sequential(spnn.Conv3d, BatchNorm, spnn.ReLU)
    ResidualBlock (and Bottleneck)                     relu(net(x) + downsample(x)),  net = Sequential(Conv3d, BN, ReLU, Conv3d, BN)
    stem                                               nn.Sequential(Conv3d, BN, ReLU, Conv3d, BN, ReLU)

where `BatchNorm` / `SyncBatchNorm` are classes the model file defines itself (nn.BatchNorm1d / nn.SyncBatchNorm through
`fapply`), the decoder concatenates with `torchsparse.cat([up(x), skip])` and the criterion is `pcseg.loss.Losses`
(CrossEntropyLoss + lovasz_softmax, R:pcseg/loss/__init__.py:106-115).

`fuse(model)` walks the module tree ONCE and swaps the `forward` of the containers it recognises BY STRUCTURE for the fused
passes of this package -- the model source, its parameters, buffers and state_dict keys stay untouched (checkpoints are
interchangeable both ways; `unfuse(model)` restores the original forwards):

  * Conv3d -> BatchNorm [-> ReLU] runs inside an nn.Sequential as: convolution whose write-back also produces the BatchNorm
    statistics (csrc/conv_common.h) -> ONE apply pass (normalise + affine [+ residual] [+ ReLU], csrc/norm.hip) and ONE backward
    pass; SyncBatchNorm layers all-reduce the (sum, sum^2, n) vector exactly like `fused.FusedBatchNorm(sync=True)`;
  * `relu(net(x) + downsample(x))`: the residual add and the final ReLU ride in the apply pass of net's last BatchNorm;
  * an up-convolution block's output is handed on as a pending tensor: `torchsparse.cat([y, skip])` then lets the apply pass
    write the concatenation (no torch.cat copy; strided dy in backward); any other use materialises it on first access;
  * `Losses.lov_loss` (the reference's per-class python loop: 19 sorts + ~25 launches per class and their autograd graph) ->
    `pcs_lovasz_softmax_f32` for device tensors, same value and gradient (tests/test_hip_parity.py::test_lovasz_softmax_*);
    `Losses.ce_loss` (plain nn.CrossEntropyLoss: torch's nll_loss reduces on ONE workgroup) -> the written-out masked mean.

  * [glue] the module-level helpers `initial_voxelize` / `voxel_to_point` / `point_to_voxel` the model file imported from its
    `utils.py` (R:pcseg/model/segmentor/voxel/minkunet/utils.py:11-105 and the identical copies next to spvcnn.py / rpvnet.py) are
    re-bound, in the MODEL module's namespace, to this package's equivalents (workloads/pointvoxel.py: unique + query + count from
    one stable sort, the whole trilinear corner map in one kernel, the level's cached hash table) -- only when their source text
    is byte-identical to the reference's (SHA-1 below), i.e. when it is known exactly what they compute;
    likewise `point_to_range` (rpvnet.py:73-91: its `torch.Tensor([w-1, h-1]).cuda()` is a device synchronisation per call) and
    `range_to_point` of rpvnet.py:31-51 (a python loop of `F.grid_sample` per frame, whose torch backward -- channel
    loops of float atomics into NCHW planes -- is 25 % of an RPVNet step) -> `rangelib.range_to_point` (csrc/rangesample.hip);
  * [forward] SPVCNN (spvcnn.py:399-456, SHA-1 of its source): the classifier over `cat([z1.F, z2.F, z3.F])` as three column blocks
    of the Linear, without the (N, 480) concatenation;
  * [forward] a model whose class is named MinkUNet and whose `forward` source is byte-identical to
    R:pcseg/model/segmentor/voxel/minkunet/minkunet.py:385-434 runs, in training mode, the same graph with the classifier applied
    on the voxels before the trilinear interpolation (`fused.devoxelized_linear`: interpolation and the Linear commute; the
    (N, 480) point-feature tensor and its gradient are never materialised) and ONE `loss.item()` instead of two. Same
    submodules, same criterion, same returned dictionaries; eval mode keeps the reference's forward.

  * [point MLPs] `nn.Sequential(nn.Linear, nn.BatchNorm1d | nn.SyncBatchNorm, nn.ReLU)` on plain (N, C) point features (the
    `point_transforms` of R:pcseg/model/segmentor/fusion/spvcnn/spvcnn.py:335-352 and rpvnet.py:571-592): the Linear's weight
    gradient x^T dy -- a (C_in, C_out) result contracted over ~2 M points, which hipBLASLt runs on a handful of workgroups (8.6 ms
    per SPVCNN step) -- on the split-reduction wgrad kernel, BatchNorm + ReLU as the fused passes.

  * [dense layers] in a model that contains sparse convolutions, every remaining stock `nn.BatchNorm1d` / `nn.SyncBatchNorm` and
    `nn.Linear` keeps its class name, parameters and state_dict keys but runs (N, C) device inputs of >= 4096 rows through the same
    fused BatchNorm passes / the split-reduction weight gradient (Cylinder_TS normalises voxel features with plain BatchNorm1d after
    every convolution, R:pcseg/model/segmentor/voxel/cylinder3d/cylinder_ts.py:103-124: torch's channels-last BatchNorm kernels are
    30 of its 170 ms step); any other input takes the stock forward. Every plain `nn.CrossEntropyLoss` child -> the masked mean.

Limits: the re-classed modules are instances of classes created at run time, so pickling a MODULE OBJECT (`torch.save(model)`) is not
supported after `fuse` -- `state_dict()` / `load_state_dict()` (what R:train.py:285-318 uses) and `copy.deepcopy` are; `unfuse(model)`
restores the original classes first if a whole-module pickle is needed.

Anything the pass does not recognise keeps its own forward; a module with forward hooks on a BatchNorm / ReLU that would be
skipped is left alone. `install_as_torchsparse(fuse=True)` applies the pass automatically the first time a model is called.
"""
import hashlib
import inspect
import os
import re
import sys

import torch
from torch import nn

from . import modules as spnn
from . import native
from .fused import _FusedBN
from .sparse import SparseTensor

__all__ = ["fuse", "unfuse", "restore_glue", "install_auto_fuse", "uninstall_auto_fuse", "PendingBatchNorm"]


# ---------------------------------------------------------------------------------------------------------------------
# BatchNorm of any nn.BatchNorm1d / nn.SyncBatchNorm subclass through the fused passes
def _bn_like(m):
    """A BatchNorm over SparseTensor features: an nn.BatchNorm1d / nn.SyncBatchNorm (sub)class with the standard state."""
    return (isinstance(m, (nn.BatchNorm1d, nn.SyncBatchNorm)) and m.affine and m.track_running_stats and
            m.momentum is not None and m.weight is not None and m.bias is not None and _world_group(m))


def _world_group(m):
    """The fused passes all-reduce SyncBatchNorm statistics over the whole job (fused._stats_group): a layer built for a
    sub-group (`convert_sync_batchnorm(model, process_group=sub)`) keeps torch's forward."""
    pg = getattr(m, "process_group", None)
    if pg is None:
        return True
    try:
        import torch.distributed as dist
        return dist.is_initialized() and (pg is dist.group.WORLD or dist.get_world_size(pg) == dist.get_world_size())
    except Exception:
        return False


def _relu_like(m):
    return isinstance(m, nn.ReLU)


def _quiet(m):
    """No forward hooks that the fused pass would skip."""
    return not (m._forward_hooks or m._forward_pre_hooks)


def _backend_fuses(feats):
    be = native.backend()
    return hasattr(be, "bn_apply") and feats.dim() == 2 and feats.dtype in (torch.float32, torch.bfloat16, torch.float16)


def bn_forward(bn, input, residual=None, relu=False, cat_with=None, in_slope=None):
    """`relu(bn(input) + residual)` [concatenated with cat_with] as one fused pass, on the parameters / buffers of the
    caller's own BatchNorm module (nn.BatchNorm1d semantics in train and eval mode; nn.SyncBatchNorm: the statistics are
    all-reduced over the default process group)."""
    x = input.feats
    r = residual.feats if isinstance(residual, SparseTensor) else residual
    tail = cat_with.feats if isinstance(cat_with, SparseTensor) else cat_with
    if tail is not None and (x.shape[1] % 4 or tail.shape[1] % 4 or tail.shape[0] != x.shape[0] or tail.dtype != x.dtype):
        y = bn_forward(bn, input, residual=residual, relu=relu)
        return y._like(torch.cat([y.feats, tail.to(y.feats.dtype)], dim=1))
    if r is not None and r.dtype != x.dtype:
        r = r.to(x.dtype)
    if bn.training:
        if bn.__dict__.get("_pcs_bumped", False):
            bn.__dict__["_pcs_bumped"] = False   # the root model's pre-forward hook already counted this step (one _foreach_add_)
        else:
            bn.num_batches_tracked.add_(1)
        pre = getattr(input, "bn_sums", None)
        if pre is not None:
            sums, of, ver = pre
            pre = sums if (of is x and x._version == ver) else None
        from . import fused as _fz
        link = _fz.BNLink() if (_fz.LINK_BN_BWD and tail is None and torch.is_grad_enabled()) else None
        y = _FusedBN.apply(x, r, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, bn.momentum, relu,
                           isinstance(bn, nn.SyncBatchNorm), input.cmaps, input.stride, pre, tail, link, in_slope)
        if link is not None and link.x is not None:
            y._pcs_bn_link = link   # the one sparse convolution that consumes y leaves this BatchNorm's backward statistics (fused.BNLink)
    else:
        inv = torch.rsqrt(bn.running_var.double() + bn.eps)
        stat = torch.cat([bn.running_mean.double(), inv]).contiguous()
        y = native.backend().bn_apply(x.contiguous(), r.contiguous() if r is not None else None, stat, bn.weight, bn.bias, relu,
                                      tail=tail.contiguous() if tail is not None else None)
    return input._like(y)


class PendingBatchNorm(SparseTensor):
    """The output of a fused up-convolution block before its BatchNorm apply pass has run. `torchsparse.cat([pending, skip])`
    lets that pass write the concatenation; reading `.feats` / `.F` (any other consumer) runs the plain pass first."""

    def __init__(self, conv_out, bn, relu):
        self._feats = None
        self._todo = (conv_out, bn, relu)
        self.coords, self.stride = conv_out.coords, conv_out.stride
        self.cmaps, self.kmaps = conv_out.cmaps, conv_out.kmaps

    def _resolve(self, cat_with=None):
        if self._todo is not None:
            conv_out, bn, relu = self._todo
            self._todo = None
            out = bn_forward(bn, conv_out, relu=relu, cat_with=cat_with)
            if cat_with is not None:
                # a later reader of THIS tensor's features (deep supervision, a hook holding the object) sees the BatchNorm
                # output: the left columns of the concatenation
                self._feats = out.feats[:, :conv_out.feats.shape[1]]
                return out
            self._feats = out.feats
        return None

    @property
    def feats(self):
        self._resolve()
        return self._feats

    @feats.setter
    def feats(self, v):
        self._todo = None
        self._feats = v

    F = property(lambda self: self.feats, lambda self, v: setattr(self, "feats", v))

    def cat_with(self, other):
        """cat([self, other]) -- fused into the apply pass when it has not run yet."""
        if self._todo is not None and isinstance(other, SparseTensor) and other.feats.shape[0] == self._todo[0].feats.shape[0]:
            return self._resolve(cat_with=other)
        return None


# ---------------------------------------------------------------------------------------------------------------------
# nn.Sequential of sparse layers
def _dense_bn(m):
    """nn.BatchNorm1d / nn.SyncBatchNorm applied to plain (N, C) tensors: the stock forward, not a SparseTensor wrapper."""
    return _bn_like(m) and type(m).forward in (nn.BatchNorm1d.forward, nn.SyncBatchNorm.forward, nn.modules.batchnorm._BatchNorm.forward)


class _Plan:
    """What a recognised nn.Sequential runs: ('cbr', conv, bn, relu_module or None) triples over SparseTensors, ('lbr', linear, bn,
    relu or None) triples over plain (N, C) tensors, and ('m', module) for the rest."""

    def __init__(self, seq):
        mods = list(seq.children())
        self.steps, i = [], 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, spnn.Conv3d) and i + 1 < len(mods) and _bn_like(mods[i + 1]) and m.bias is None:
                act = mods[i + 2] if i + 2 < len(mods) and _relu_like(mods[i + 2]) else None
                self.steps.append(("cbr", m, mods[i + 1], act))
                i += 3 if act is not None else 2
            elif type(m) is nn.Linear and i + 1 < len(mods) and _dense_bn(mods[i + 1]) and mods[i + 1].num_features == m.out_features:
                act = mods[i + 2] if i + 2 < len(mods) and type(mods[i + 2]) is nn.ReLU else None
                self.steps.append(("lbr", m, mods[i + 1], act))
                i += 3 if act is not None else 2
            else:
                self.steps.append(("m", m))
                i += 1
        self.n_fused = sum(1 for s in self.steps if s[0] in ("cbr", "lbr"))
        self.n_dense = sum(1 for s in self.steps if s[0] == "lbr")
        last = self.steps[-1] if self.steps else None
        self.ends_in_bn = last is not None and last[0] == "cbr" and last[3] is None     # residual + final ReLU can ride here
        self.pending = last is not None and last[0] == "cbr" and last[1].transposed    # decoder up-conv: cat may follow


_POINT_MAPS = {}   # identity maps (row count, device) of the point MLPs' weight gradients: one per step, shared by all of them


class _PointLinear(torch.autograd.Function):
    """y = x W^T + b over ~2 M point rows: forward and input gradient stay dense GEMMs (hipBLASLt through torch), the weight gradient
    x^T dy -- (C_in, C_out), contracted over every point -- runs on the split-reduction wgrad kernel (as functional._PointwiseConv)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        from .functional import _amp_dtype
        hd = _amp_dtype(x)
        ctx.hd, ctx.has_bias, ctx.in_dtype = hd, bias is not None, x.dtype
        if hd is None:
            ctx.save_for_backward(x, weight)
            return torch.nn.functional.linear(x.float(), weight.float(), bias.float() if bias is not None else None)
        xh = x.to(hd)   # kept for the weight gradient: one conversion of the (N, C_in) rows instead of one per direction
        ctx.save_for_backward(xh, weight)
        return torch.nn.functional.linear(xh, weight.to(hd), bias.to(hd) if bias is not None else None)

    @staticmethod
    def backward(ctx, dy):
        from .functional import _identity_map
        x, weight = ctx.saved_tensors
        dy = dy.contiguous()
        hd = ctx.hd
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = (dy.float().matmul(weight.float()) if hd is None else dy.to(hd).matmul(weight.to(hd))).to(ctx.in_dtype)
        if ctx.needs_input_grad[1]:
            if len(_POINT_MAPS) > 4:
                _POINT_MAPS.clear()
            km = _identity_map(x.shape[0], x.device, _POINT_MAPS)
            be = native.backend()
            xa, cin = x.contiguous(), x.shape[1]
            if cin % 4:   # e.g. Cylinder_TS's first point layer, Linear(9, 64): zero columns bring the rows to 16-byte granularity
                xa = torch.nn.functional.pad(xa, (0, (-cin) % 4))
            if hd is not None:
                gw = be.conv_wgrad_h(xa.to(hd), dy.to(hd), km, 0)[0]
            else:
                gw = be.conv_wgrad(xa.float(), dy.float(), km, 0)[0]
            gw = gw[:cin].t().to(weight.dtype)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = dy.float().sum(0).to(weight.dtype)
        return gx, gw, gb


def _dense_step(lin, bn, act, x):
    """Linear -> BatchNorm -> [ReLU] on a plain (N, C) tensor through the fused passes; the modules themselves where they do not apply."""
    if not (_rows_ok(x) and _quiet(lin) and _quiet(bn) and (act is None or _quiet(act))):
        h = bn(lin(x))
        return act(h) if act is not None else h
    if lin.out_features % 4 == 0:
        h = _PointLinear.apply(x, lin.weight, lin.bias)
    else:
        h = lin(x)
    return _dense_bn_apply(bn, h, act is not None)


def _dense_bn_apply(bn, h, relu):
    if bn.training:
        if bn.__dict__.get("_pcs_bumped", False):
            bn.__dict__["_pcs_bumped"] = False
        else:
            bn.num_batches_tracked.add_(1)
        return _FusedBN.apply(h, None, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, bn.momentum, relu,
                              isinstance(bn, nn.SyncBatchNorm), None, None, None, None)
    inv = torch.rsqrt(bn.running_var.double() + bn.eps)
    stat = torch.cat([bn.running_mean.double(), inv]).contiguous()
    return native.backend().bn_apply(h.contiguous(), None, stat, bn.weight, bn.bias, relu)


def _rows_ok(x):
    return (isinstance(x, torch.Tensor) and x.dim() == 2 and x.is_cuda and x.shape[0] >= 4096 and
            x.dtype in (torch.float32, torch.bfloat16, torch.float16) and hasattr(native.backend(), "bn_apply"))


def _dense_bn_forward(self, x):
    """forward of a stock nn.BatchNorm1d / nn.SyncBatchNorm re-classed by fuse(): (N, C) device rows through the fused passes."""
    if not (_rows_ok(x) and _quiet(self)):
        return self.__dict__["_pcs_orig_class"].forward(self, x)
    return _dense_bn_apply(self, x, False)


def _dense_linear_forward(self, x):
    """forward of a stock nn.Linear re-classed by fuse(): tall (N, C_in) device inputs get the split-reduction weight gradient."""
    if not (_rows_ok(x) and _quiet(self) and self.out_features % 4 == 0):
        return self.__dict__["_pcs_orig_class"].forward(self, x)
    return _PointLinear.apply(x, self.weight, self.bias)


def _skip_through(conv, x):
    """(conv(x), x routed through conv's autograd node) when the block's input also feeds its skip path: the dgrad kernel then adds
    the skip gradient in its write-back instead of autograd summing two tensors (functional._SparseConv, with_skip) [r6]."""
    if (os.environ.get("PCS_SKIP_FUSED", "1") != "0" and type(conv) is spnn.Conv3d and _quiet(conv) and torch.is_grad_enabled() and
            isinstance(x, SparseTensor) and x.feats.requires_grad and tuple(conv.kernel_size) != (1, 1, 1)):
        return conv(x, with_skip=True)
    return conv(x), x


def _run_plan(seq, plan, x, residual=None, final_relu=False):
    """residual: a SparseTensor, or a callable xs -> SparseTensor that is handed the block input AFTER the first convolution ran
    (so that the input can be routed through that convolution's autograd node, _skip_through)."""
    n = len(plan.steps)
    if callable(residual) and not (n and plan.steps[0][0] == "cbr"):
        residual = residual(x)
    for j, st in enumerate(plan.steps):
        if st[0] == "m":
            x = st[1](x)
            continue
        if st[0] == "lbr":
            if isinstance(x, torch.Tensor):
                x = _dense_step(st[1], st[2], st[3], x)
            else:
                x = st[2](st[1](x))
                x = st[3](x) if st[3] is not None else x
            continue
        _, conv, bn, act = st
        if j == 0 and callable(residual):
            h, xs = _skip_through(conv, x)
            residual = residual(xs)
        else:
            h = conv(x)   # the module call (its hooks run); emit_bn_stats makes the write-back leave the BatchNorm statistics
        if not (isinstance(h, SparseTensor) and _backend_fuses(h.feats) and _quiet(bn) and (act is None or _quiet(act))):
            h = bn(h)
            x = act(h) if act is not None else h
            if j == n - 1 and residual is not None:
                x = x + residual
                x = x._like(torch.relu(x.feats)) if final_relu else x
            continue
        last = j == n - 1
        if last and residual is not None:
            x = bn_forward(bn, h, residual=residual, relu=final_relu or act is not None)
        elif last and plan.pending and residual is None and os.environ.get("PCS_CAT_FUSED", "1") != "0":
            x = PendingBatchNorm(h, bn, act is not None)
        else:
            x = bn_forward(bn, h, relu=act is not None or (last and final_relu))
    return x


def _sequential_forward(self, input):
    """forward of a recognised nn.Sequential (class-level: the module is re-classed, so copy.deepcopy keeps working)."""
    plan = self.__dict__["_pcs_plan"]
    if not isinstance(input, SparseTensor) and not (plan.n_dense and isinstance(input, torch.Tensor)):
        return nn.Sequential.forward(self, input)
    return _run_plan(self, plan, input)


_RESIDUAL_SRC = re.compile(r"self\.relu\(\s*self\.net\(x\)\s*\+\s*self\.downsample\(x\)\s*\)")


def _residual_block(m):
    """`out = self.relu(self.net(x) + self.downsample(x))` (R:pcseg/model/segmentor/voxel/minkunet/minkunet.py:83-129, also the
    Bottleneck :132-186): recognised by its attributes AND by the source of its forward."""
    net, ds, act = getattr(m, "net", None), getattr(m, "downsample", None), getattr(m, "relu", None)
    if not (isinstance(net, nn.Sequential) and _relu_like(act) and isinstance(ds, (nn.Identity, nn.Sequential))):
        return None
    try:
        src = inspect.getsource(type(m).forward)
    except (OSError, TypeError):
        return None
    if not _RESIDUAL_SRC.search(src) or src.count("self.") != 3:
        return None
    plan = _Plan(net)
    if not plan.ends_in_bn:
        return None
    ds_plan = None
    if isinstance(ds, nn.Sequential):
        ds_plan = _Plan(ds)
    return plan, ds_plan


def _residual_forward(self, x):
    plan, ds_plan = self.__dict__["_pcs_plan"]
    if not (isinstance(x, SparseTensor) and _quiet(self.relu) and _quiet(self.net) and _quiet(self.downsample)):
        return self.__dict__["_pcs_orig_class"].forward(self, x)
    ds = self.downsample
    return _run_plan(self.net, plan, x, residual=(lambda xs: xs if ds_plan is None else _run_plan(ds, ds_plan, xs)), final_relu=True)


_FUSED_CLASSES = {}


def _reclass(m, forward, plan, undo):
    """Give `m` a subclass of its own class whose forward is the fused one (cached per class; same name and module)."""
    cls = type(m)
    sub = _FUSED_CLASSES.get((cls, forward))
    if sub is None:
        sub = type(cls.__name__, (cls,), {"forward": forward, "__module__": cls.__module__, "_pcs_fused_class": True})
        sub.__qualname__ = cls.__qualname__
        _FUSED_CLASSES[(cls, forward)] = sub
    m.__dict__["_pcs_plan"], m.__dict__["_pcs_orig_class"] = plan, cls
    m.__class__ = sub
    undo.append(("class", m, cls))


# ---------------------------------------------------------------------------------------------------------------------
# the criterion
class _MaskedCE(nn.Module):
    """nn.CrossEntropyLoss(ignore_index, label_smoothing, weight=None, reduction='mean') on (N, C) logits, written out as a
    log-softmax + gather + masked mean (same value; torch's nll_loss forward / backward reduce on one workgroup: 1.3 ms each
    for 1.2 M rows). Holds no state: swapping it in changes no state_dict."""

    def __init__(self, ce):
        super().__init__()
        self.ignore_index, self.label_smoothing, self._orig = ce.ignore_index, float(ce.label_smoothing), [ce]

    def forward(self, logits, target):
        if not (logits.is_cuda and logits.dim() == 2 and target.dim() == 1 and not target.is_floating_point()):
            return self._orig[0](logits, target)
        logp = torch.nn.functional.log_softmax(logits.float(), dim=1)
        keep = target != self.ignore_index
        picked = logp.gather(1, target.clamp(0, logits.shape[1] - 1).unsqueeze(1)).squeeze(1)
        per_row = -(1.0 - self.label_smoothing) * picked
        if self.label_smoothing > 0:
            per_row = per_row - self.label_smoothing * logp.mean(dim=1)
        return (per_row * keep).sum() / keep.sum()


def _lovasz_router(orig):
    from .workloads.losses import lovasz_softmax_device

    def lovasz_softmax(probas, labels, classes="present", per_image=False, ignore=None):
        ok = (isinstance(probas, torch.Tensor) and probas.is_cuda and probas.dim() == 2 and classes == "present" and
              not per_image and probas.shape[1] <= 60 and labels.dim() == 1 and not labels.is_floating_point())
        if not ok:
            return orig(probas, labels, classes=classes, per_image=per_image, ignore=ignore)
        return lovasz_softmax_device(probas.float(), labels.long(), ignore=ignore)
    lovasz_softmax._pcs_orig = orig
    return lovasz_softmax


def _fuse_criterion(m, undo):
    n = 0
    lov = m.__dict__.get("lov_loss")
    if (callable(lov) and getattr(lov, "__name__", "") == "lovasz_softmax" and not hasattr(lov, "_pcs_orig") and
            getattr(lov, "__module__", "").endswith("lovasz_losses")):
        m.lov_loss = _lovasz_router(lov)
        undo.append(("attr", m, "lov_loss", lov))
        n += 1
    ce = m._modules.get("ce_loss")
    if type(ce) is nn.CrossEntropyLoss and ce.weight is None and ce.reduction == "mean" and _quiet(ce):
        m._modules["ce_loss"] = _MaskedCE(ce)
        undo.append(("submodule", m, "ce_loss", ce))
        n += 1
    return n


def _bump(module, args):
    """Root pre-forward hook: num_batches_tracked of every fused BatchNorm in one _foreach_add_ per training step instead of one
    scalar kernel per layer; a layer that is not reached in a forward is corrected when the next forward starts."""
    st = module.__dict__.get("_pcs_fused")
    if st is None:
        return
    bns = st["bns"]
    if module.training and bns and all(b.training for b in bns):
        for b in bns:
            if b.__dict__.get("_pcs_bumped", False):   # counted last step but never ran: take that count back
                b.num_batches_tracked.sub_(1)
            b.__dict__["_pcs_bumped"] = True
        torch._foreach_add_([b.num_batches_tracked for b in bns], 1)


# ---------------------------------------------------------------------------------------------------------------------
# the point <-> voxel glue and the MinkUNet forward, recognised by the SHA-1 of their source text (computed from /root/reference)
_GLUE_SHA1 = {"initial_voxelize": "d86d7900a8d70be1746f57eb311959ae0e23770a",
              "voxel_to_point": "ab9c4461fe068ac54a5105b96b90072206bb1f82",
              "point_to_voxel": "546ed12789757812f69502843a6d5bd4216130af"}
_MINKUNET_FORWARD_SHA1 = "e5ef7d830c649b15021924421d95424e7a0501c8"
_SPVCNN_FORWARD_SHA1 = "905bca0f3105c02f8d55f691f6d266f5a8558d98"     # R:pcseg/model/segmentor/fusion/spvcnn/spvcnn.py:399-456
# R:pcseg/model/segmentor/fusion/rpvnet/rpvnet.py:31-51: range_to_point = resample_grid_stacked = grid_sample per frame
_RANGE_TO_POINT_SHA1 = {"range_to_point": "84c1b2d60560e333892da86ed6372b6a55585c61",
                        "resample_grid_stacked": "c0f10770b7114ddf6cd1193e721fea602d6b905f"}
_POINT_TO_RANGE_SHA1 = "749449227dbd17dc3db119ac55bef79a4e68ac59"   # rpvnet.py:73-91


def _source_sha1(fn):
    try:
        return hashlib.sha1(inspect.getsource(fn).encode()).hexdigest()
    except (OSError, TypeError):
        return None


_GLUE_ORIG = {}   # (module name, helper name) -> the reference's function (process-wide: `restore_glue()` puts them back)


def restore_glue():
    """Undo every helper re-binding `fuse(..., glue=True)` made in this process."""
    for (modname, name), fn in list(_GLUE_ORIG.items()):
        ns = sys.modules.get(modname)
        if ns is not None:
            setattr(ns, name, fn)
        del _GLUE_ORIG[(modname, name)]


def _range_to_point_via(orig):
    from .rangelib import range_to_point as fused

    def range_to_point(feature_map, pxpy, grid_sample_mode="bilinear"):
        # one launch each way instead of grid_sample per frame (csrc/rangesample.hip); what the kernels do not serve goes to the
        # reference's own function
        return fused(feature_map, pxpy, grid_sample_mode, fallback=orig)
    range_to_point.__module__ = "openpcseg_amd.rangelib"
    return range_to_point


def _fuse_glue(model):
    """Re-bind the reference's glue helpers in the namespaces of the modules that define this model's classes. The namespace
    belongs to the model FILE, so the re-binding serves every instance built from it (same results; `restore_glue()` undoes it)."""
    from .workloads import pointvoxel as pv
    n, seen = 0, set()
    for m in model.modules():
        modname = type(m).__dict__.get("__module__", type(m).__module__)
        if modname in seen:
            continue
        seen.add(modname)
        ns = sys.modules.get(modname)
        if ns is None:
            continue
        for name, sha in _GLUE_SHA1.items():
            fn = ns.__dict__.get(name)
            if fn is None or getattr(fn, "__module__", "").startswith("openpcseg_amd") or not inspect.isfunction(fn):
                continue
            if _source_sha1(fn) == sha:
                setattr(ns, name, getattr(pv, name))
                _GLUE_ORIG[(modname, name)] = fn
                n += 1
            else:
                _warn_unrecognised("%s.%s" % (modname, name), "its source text differs from the reference's")
        r2p = ns.__dict__.get("range_to_point")
        if (inspect.isfunction(r2p) and not getattr(r2p, "__module__", "").startswith("openpcseg_amd") and
                all(_source_sha1(ns.__dict__.get(k)) == v for k, v in _RANGE_TO_POINT_SHA1.items())):
            setattr(ns, "range_to_point", _range_to_point_via(r2p))
            _GLUE_ORIG[(modname, "range_to_point")] = r2p
            n += 1
        p2r = ns.__dict__.get("point_to_range")
        if (inspect.isfunction(p2r) and not getattr(p2r, "__module__", "").startswith("openpcseg_amd") and
                _source_sha1(p2r) == _POINT_TO_RANGE_SHA1):
            from .rangelib import point_to_range
            setattr(ns, "point_to_range", point_to_range)   # no host-to-device copy (= device synchronisation) per call
            _GLUE_ORIG[(modname, "point_to_range")] = p2r
            n += 1
    return n


def _prebuild_encoder(model, x0):
    """functional.prebuild_coords for the reference's encoder layout: stage1..stage4, each opened by a strided, non-transposed
    Conv3d (R:pcseg/model/segmentor/voxel/minkunet/minkunet.py:233-262) whose own stride / kernel size are read here. Any other
    layout: nothing is prebuilt. The coordinates are the ones those convolutions would compute themselves (same call, same dict)."""
    steps = model.__dict__.get("_pcs_enc_steps")
    if steps is None:
        from .modules import Conv3d
        steps = []
        for name in ("stage1", "stage2", "stage3", "stage4"):
            st = getattr(model, name, None)
            conv = next((m for m in st.modules() if isinstance(m, Conv3d)), None) if isinstance(st, nn.Module) else None
            if conv is None or conv.transposed or all(int(v) == 1 for v in conv.stride):
                steps = []
                break
            steps.append((tuple(conv.stride), tuple(conv.kernel_size)))
        model.__dict__["_pcs_enc_steps"] = steps
    if steps and os.environ.get("PCS_PREBUILD_LEVELS", "1") != "0":
        from . import functional as F
        F.prebuild_coords(x0, steps)
    return x0


def _minkunet_forward(self, batch_dict, return_logit=False, return_tta=False):
    """Training-mode forward of the reference's MinkUNet (R:pcseg/model/segmentor/voxel/minkunet/minkunet.py:385-434) with the
    classifier commuted in front of the three trilinear interpolations; everything else line by line as the reference."""
    from . import sparse as ts
    from .fused import devoxelized_linear
    from .workloads.pointvoxel import initial_voxelize, point_maps, voxel_to_point
    lin = self.classifier[0] if isinstance(self.classifier, nn.Sequential) and len(self.classifier) == 1 else None
    x = batch_dict["lidar"]
    if not (self.training and isinstance(lin, nn.Linear) and x.F.is_cuda and _quiet(lin) and _quiet(self.classifier) and
            _quiet(self.dropout) and hasattr(native.backend(), "corner_map") and os.environ.get("PCS_CLASSIFIER_COMMUTE", "1") != "0"):
        return self.__dict__["_pcs_orig_class"].forward(self, batch_dict, return_logit, return_tta)
    x.F = x.F[:, :self.in_feature_dim]
    z = ts.PointTensor(x.F, x.C.float())
    x0 = _prebuild_encoder(self, initial_voxelize(z, self.pres, self.vres))
    x0 = self.stem(x0)
    z0 = voxel_to_point(x0, z, nearest=False)
    x1 = self.stage1(x0)
    x2 = self.stage2(x1)
    x3 = self.stage3(x2)
    x4 = self.stage4(x3)
    drop = self.dropout
    t1 = devoxelized_linear(lin.weight, 0, x4.F, *point_maps(x4, z0), x4.kmaps)          # = voxel_to_point(x4, z0).F @ W[:, :c4]^T
    c4 = x4.F.shape[1]
    x4.F = torch.nn.functional.dropout(x4.F, drop.p, drop.training, False)                # out of place: x4.F feeds t1's weight gradient
    y1 = self.up1[0](x4)
    y1 = ts.cat([y1, x3])
    y1 = self.up1[1](y1)
    y2 = self.up2[0](y1)
    y2 = ts.cat([y2, x2])
    y2 = self.up2[1](y2)
    t2 = devoxelized_linear(lin.weight, c4, y2.F, *point_maps(y2, z0), y2.kmaps)
    c2 = y2.F.shape[1]
    y2.F = torch.nn.functional.dropout(y2.F, drop.p, drop.training, False)
    y3 = self.up3[0](y2)
    y3 = ts.cat([y3, x1])
    y3 = self.up3[1](y3)
    y4 = self.up4[0](y3)
    y4 = ts.cat([y4, x0])
    y4 = self.up4[1](y4)
    t3 = devoxelized_linear(lin.weight, c4 + c2, y4.F, *point_maps(y4, z0), y4.kmaps)
    out = t1 + t2 + t3
    if lin.bias is not None:
        out = out + lin.bias
    target = batch_dict["targets"].F.long().cuda(non_blocking=True)
    coords_xyz = batch_dict["lidar"].C[:, :3].float()
    loss = self.criterion_losses(out, target, xyz=coords_xyz, offset=batch_dict["offset"])
    value = loss.item()
    return {"loss": loss}, {"loss": value}, {"loss": value}


def _spvcnn_forward(self, batch_dict, return_logit=False, return_tta=False):
    """Training-mode forward of the reference's SPVCNN (R:pcseg/model/segmentor/fusion/spvcnn/spvcnn.py:399-456), line by line, with
    ONE change: `classifier(torch.cat([z1.F, z2.F, z3.F], 1))` is evaluated as the sum of the three column blocks of the Linear
    (`fused._SkinnyLinearParts`): the (N, 480) concatenation of 1.9 M point rows -- 3.7 GB written and read back, and sliced again in
    backward -- is never built. Eval mode, other `multi_scale` settings and hooked classifiers take the reference's forward."""
    from . import sparse as ts
    from .fused import _SkinnyLinearParts
    from .workloads.pointvoxel import initial_voxelize, point_to_voxel, voxel_to_point
    lin = self.classifier[0] if isinstance(self.classifier, nn.Sequential) and len(self.classifier) == 1 else None
    x = batch_dict["lidar"]
    if not (self.training and isinstance(lin, nn.Linear) and x.F.is_cuda and getattr(self, "multi_scale", None) == "concat" and
            _quiet(lin) and _quiet(self.classifier) and lin.out_features % 4 == 0 and hasattr(native.backend(), "corner_map") and
            os.environ.get("PCS_CLASSIFIER_PARTS", "1") != "0"):
        return self.__dict__["_pcs_orig_class"].forward(self, batch_dict, return_logit, return_tta)
    x.F = x.F[:, :self.in_feature_dim]
    z = ts.PointTensor(x.F, x.C.float())
    x0 = _prebuild_encoder(self, initial_voxelize(z, self.pres, self.vres))
    x0 = self.stem(x0)
    z0 = voxel_to_point(x0, z, nearest=False)
    x1 = point_to_voxel(x0, z0)
    x1 = self.stage1(x1)
    x2 = self.stage2(x1)
    x3 = self.stage3(x2)
    x4 = self.stage4(x3)
    z1 = voxel_to_point(x4, z0)
    z1.F = z1.F + self.point_transforms[0](z0.F)
    y1 = point_to_voxel(x4, z1)
    y1.F = self.dropout(y1.F)
    y1 = self.up1[0](y1)
    y1 = ts.cat([y1, x3])
    y1 = self.up1[1](y1)
    y2 = self.up2[0](y1)
    y2 = ts.cat([y2, x2])
    y2 = self.up2[1](y2)
    z2 = voxel_to_point(y2, z1)
    z2.F = z2.F + self.point_transforms[1](z1.F)
    y3 = point_to_voxel(y2, z2)
    y3.F = self.dropout(y3.F)
    y3 = self.up3[0](y3)
    y3 = ts.cat([y3, x1])
    y3 = self.up3[1](y3)
    y4 = self.up4[0](y3)
    y4 = ts.cat([y4, x0])
    y4 = self.up4[1](y4)
    z3 = voxel_to_point(y4, z2)
    z3.F = z3.F + self.point_transforms[2](z2.F)
    parts = [z1.F, z2.F, z3.F]
    if (all(p.dim() == 2 and p.shape[1] % 4 == 0 and p.dtype in (torch.float32, torch.bfloat16, torch.float16) for p in parts) and
            parts[0].shape[0] >= 4096 and sum(p.shape[1] for p in parts) == lin.in_features):
        hd = next((p.dtype for p in parts if p.dtype != torch.float32), None)
        if hd is None and torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") in (torch.bfloat16, torch.float16):
            hd = torch.get_autocast_dtype("cuda")
        if len(_POINT_MAPS) > 4:
            _POINT_MAPS.clear()
        out = _SkinnyLinearParts.apply(lin.weight, lin.bias, _POINT_MAPS, hd, *parts)
    else:
        out = self.classifier(torch.cat(parts, dim=1))
    target = batch_dict["targets"].F.long().cuda(non_blocking=True)
    coords_xyz = batch_dict["lidar"].C[:, :3].float()
    loss = self.criterion_losses(out, target, xyz=coords_xyz, offset=batch_dict["offset"])
    value = loss.item()
    return {"loss": loss}, {"loss": value}, {"loss": value}


def _glue_is_ours(m):
    """The replacement forwards call this package's point <-> voxel helpers directly. That is only the model's own function when
    every helper the model FILE binds is byte for byte the reference's (then `_fuse_glue` has re-bound it to this package's) --
    a fork with an edited utils.py keeps its forward (and its helpers)."""
    modname = type(m).__dict__.get("__module__", type(m).__module__)
    ns = sys.modules.get(modname)
    if ns is None:
        return False
    ours = 0
    for name in _GLUE_SHA1:
        fn = ns.__dict__.get(name)
        if fn is None:
            continue          # the model file does not bind (= does not call) this helper
        if not getattr(fn, "__module__", "").startswith("openpcseg_amd"):
            return False
        ours += 1
    return ours >= 2          # initial_voxelize and voxel_to_point at least


def _warn_unrecognised(what, why):
    import warnings
    warnings.warn("openpcseg_amd.fuse: %s is not fused -- %s. The model runs on its own (unfused) code for that part: same results, "
                  "lower speed (INTEGRATION.md section 4)." % (what, why), RuntimeWarning, stacklevel=3)


# ---------------------------------------------------------------------------------------------------------------------
# Cylinder_TS blocks (R:pcseg/model/segmentor/voxel/cylinder3d/cylinder_ts.py:88-330): chains of
#     t = conv(x); t.F = LeakyReLU(t.F); t.F = BatchNorm1d(t.F)
# recognised by the SHA-1 of each block's forward [r6]. Per link of the chain the reference runs conv, leaky_relu (read + write),
# batch_norm (statistics read, apply read + write) and, backward, batch_norm_backward (2 + 3 passes) and leaky_relu_backward
# (2 reads + 1 write); fused: the convolution's write-back applies the LeakyReLU and leaves the BatchNorm statistics, ONE apply
# pass, and the backward apply pass multiplies the activation's derivative in. The blocks' residual sums ride in the last apply
# pass; the two convolutions that read the block input share one autograd node for its gradient (_skip_through).
_CYL_FORWARD_SHA1 = {"ResContextBlock": "ca01095d526ca22e88e3934d5d3c584bcc797798",
                     "ResBlock": "5e33c3d957290051e87718bb5cb51fbcdb9328c9",
                     "UpBlock": "9ec2fd9d3f0af222467bede0edc24312e70c82f6"}


def _cab(conv, act, bn, x, residual=None, want_skip=False):
    """conv -> LeakyReLU -> BatchNorm (+ residual) on a SparseTensor; -> (result, x routed through conv when want_skip)."""
    from . import functional as Fn
    ok = (type(conv) is spnn.Conv3d and type(act) is nn.LeakyReLU and _bn_like(bn) and _quiet(conv) and _quiet(act) and _quiet(bn) and
          isinstance(x, SparseTensor) and _backend_fuses(x.feats) and conv.bias is None and conv.kernel.dim() == 3 and
          Fn.conv_act_fusable(x.feats, conv.kernel) and os.environ.get("PCS_CYL_FUSED", "1") != "0")
    if not ok:
        h = conv(x)
        h.F = act(h.F)
        h.F = bn(h.F)
        if residual is not None:
            h.F = h.F + residual.F
        return (h, x) if want_skip else h
    skip = want_skip and torch.is_grad_enabled() and x.feats.requires_grad and os.environ.get("PCS_SKIP_FUSED", "1") != "0"
    out = Fn.conv3d(x, conv.kernel, kernel_size=conv.kernel_size, stride=conv.stride, dilation=conv.dilation,
                    transposed=conv.transposed, bn_stats=bn.training, with_skip=skip, act_slope=act.negative_slope)
    h, xs = out if skip else (out, x)
    y = bn_forward(bn, h, residual=residual, relu=False, in_slope=act.negative_slope)
    return (y, xs) if want_skip else y


def _cyl_context_forward(self, x):        # ResContextBlock.forward, cylinder_ts.py:137-155
    if not isinstance(x, SparseTensor):
        return self.__dict__["_pcs_orig_class"].forward(self, x)
    shortcut, xs = _cab(self.conv1, self.act1, self.bn0, x, want_skip=True)
    shortcut = _cab(self.conv1_2, self.act1_2, self.bn0_2, shortcut)
    res_a = _cab(self.conv2, self.act2, self.bn1, xs)
    return _cab(self.conv3, self.act3, self.bn2, res_a, residual=shortcut)


def _cyl_resblock_forward(self, x):       # ResBlock.forward, cylinder_ts.py:229-252
    if not isinstance(x, SparseTensor):
        return self.__dict__["_pcs_orig_class"].forward(self, x)
    res_a = _cyl_context_forward(self, x)
    if self.pooling:
        return self.pool(res_a), res_a
    return res_a


def _cyl_upblock_forward(self, x, skip):  # UpBlock.forward, cylinder_ts.py:312-330
    if not isinstance(x, SparseTensor):
        return self.__dict__["_pcs_orig_class"].forward(self, x, skip)
    up_a = _cab(self.trans_dilao, self.trans_act, self.trans_bn, x)
    up_a = self.up_subm(up_a)
    up_a.F = up_a.F + skip.F
    up_e = _cab(self.conv1, self.act1, self.bn1, up_a)
    up_e = _cab(self.conv2, self.act2, self.bn2, up_e)
    return _cab(self.conv3, self.act3, self.bn3, up_e)


_CYL_FORWARDS = {"ResContextBlock": (_cyl_context_forward, ("conv1", "act1", "bn0", "conv1_2", "act1_2", "bn0_2", "conv2", "act2", "bn1", "conv3", "act3", "bn2")),
                 "ResBlock": (_cyl_resblock_forward, ("conv1", "act1", "bn0", "conv1_2", "act1_2", "bn0_2", "conv2", "act2", "bn1", "conv3", "act3", "bn2", "pooling")),
                 "UpBlock": (_cyl_upblock_forward, ("trans_dilao", "trans_act", "trans_bn", "up_subm", "conv1", "act1", "bn1", "conv2", "act2", "bn2", "conv3", "act3", "bn3"))}


def _fuse_cylinder_block(m, undo):
    name = type(m).__name__
    if name not in _CYL_FORWARDS or getattr(type(m), "_pcs_fused_class", False):
        return 0
    fwd, attrs = _CYL_FORWARDS[name]
    if not all(hasattr(m, a) for a in attrs) or not all(type(getattr(m, a)) is spnn.Conv3d for a in attrs if a.startswith(("conv", "trans_dilao"))):
        return 0   # another model's block of the same name (the range branch's ResBlock is dense convolutions)
    if _source_sha1(type(m).forward) != _CYL_FORWARD_SHA1[name]:
        _warn_unrecognised("%s.forward" % name, "its source text differs from the reference's (recognition is by SHA-1 of the text)")
        return 0
    _reclass(m, fwd, None, undo)
    return 1


def _fuse_model_forward(m, undo):
    name = type(m).__name__
    if name in ("SPVCNN", "MinkUNet") and not getattr(type(m), "_pcs_fused_class", False):
        sha = _source_sha1(type(m).forward)
        if sha != (_SPVCNN_FORWARD_SHA1 if name == "SPVCNN" else _MINKUNET_FORWARD_SHA1):
            _warn_unrecognised("%s.forward" % name, "its source text differs from the reference's (recognition is by SHA-1 of the text)")
            return 0
        if not _glue_is_ours(m):
            _warn_unrecognised("%s.forward" % name, "a point <-> voxel helper of its model file is not the reference's")
            return 0
    if (type(m).__name__ == "SPVCNN" and not getattr(type(m), "_pcs_fused_class", False) and
            _source_sha1(type(m).forward) == _SPVCNN_FORWARD_SHA1 and
            all(hasattr(m, a) for a in ("stem", "stage1", "stage4", "up1", "up4", "classifier", "dropout", "criterion_losses",
                                        "point_transforms", "in_feature_dim", "pres", "vres", "multi_scale"))):
        _reclass(m, _spvcnn_forward, None, undo)
        return 1
    if (type(m).__name__ == "MinkUNet" and not getattr(type(m), "_pcs_fused_class", False) and
            _source_sha1(type(m).forward) == _MINKUNET_FORWARD_SHA1 and
            all(hasattr(m, a) for a in ("stem", "stage1", "stage4", "up1", "up4", "classifier", "dropout", "criterion_losses",
                                        "in_feature_dim", "pres", "vres"))):
        _reclass(m, _minkunet_forward, None, undo)
        return 1
    return 0


def _unbump(module, args, output):
    """Root forward hook: a fused BatchNorm that did NOT run in this forward (a branch not taken, a head used only in eval) gives
    the count of `_bump` back, so the counters are exact whenever the forward has returned."""
    st = module.__dict__.get("_pcs_fused")
    if st is not None:
        for b in st["bns"]:
            if b.__dict__.get("_pcs_bumped", False):
                b.num_batches_tracked.sub_(1)
                b.__dict__["_pcs_bumped"] = False


# ---------------------------------------------------------------------------------------------------------------------
def _adopt(plan, bns):
    for st in plan.steps:
        if st[0] == "cbr":
            st[1].emit_bn_stats = True
        if st[0] in ("cbr", "lbr"):
            bns.append(st[2])
    return plan.n_fused


def fuse(model, criterion=True, glue=True, forward=True):
    """Swap the forwards of the blocks this pass recognises (see the module docstring). Idempotent. Returns a dict of counts:
    {"sequential": .., "residual": .., "conv_bn": .., "criterion": .., "glue": .., "forward": .., "dense": .., "cylinder": ..}."""
    if model.__dict__.get("_pcs_fused") is not None:
        return dict(model.__dict__["_pcs_fused"]["counts"])
    counts = {"sequential": 0, "residual": 0, "conv_bn": 0, "criterion": 0, "glue": 0, "forward": 0, "dense": 0, "cylinder": 0}
    undo, bns, owned = [], [], set()
    mods = list(model.modules())
    for m in mods:
        m.__dict__["_pcs_fuse_seen"] = True
    for m in mods:
        if isinstance(m, nn.Sequential) or getattr(type(m), "_pcs_fused_class", False):
            continue
        got = _residual_block(m)
        if got is None:
            continue
        _reclass(m, _residual_forward, got, undo)
        counts["residual"] += 1
        for p, owner in ((got[0], m.net), (got[1], m.downsample)):
            if p is not None:
                counts["conv_bn"] += _adopt(p, bns)
                owned.add(id(owner))
    for m in mods:
        if isinstance(m, nn.Sequential) and type(m).forward is nn.Sequential.forward:
            plan = _Plan(m)
            if plan.n_fused == 0:
                continue
            _reclass(m, _sequential_forward, plan, undo)   # (a residual block's net too: serves the block's unfused fallback)
            if id(m) not in owned:
                counts["sequential"] += 1
                counts["conv_bn"] += _adopt(plan, bns)
        elif criterion and not isinstance(m, nn.Sequential):
            counts["criterion"] += _fuse_criterion(m, undo)
    cyl = [m for m in mods if type(m).__name__ in _CYL_FORWARDS and type(getattr(m, "conv1", None)) is spnn.Conv3d]
    if bns or cyl:
        # stock BatchNorm1d / SyncBatchNorm / Linear modules outside every recognised plan (Cylinder_TS's per-convolution BatchNorm1d on
        # `.F`, the first / last layers of its point MLP, classifiers) and plain CrossEntropyLoss children
        planned = set()
        for m in mods:
            pl = m.__dict__.get("_pcs_plan")
            for p_ in (pl if isinstance(pl, tuple) else (pl,)):
                if isinstance(p_, _Plan):
                    planned.update(id(x) for st in p_.steps for x in st[1:] if isinstance(x, nn.Module))
        for m in mods:
            if id(m) in planned or getattr(type(m), "_pcs_fused_class", False):
                continue
            if _dense_bn(m) and type(m) in (nn.BatchNorm1d, nn.SyncBatchNorm):
                _reclass(m, _dense_bn_forward, None, undo)
                bns.append(m)
                counts["dense"] += 1
            elif type(m) is nn.Linear:
                _reclass(m, _dense_linear_forward, None, undo)
                counts["dense"] += 1
            elif criterion:
                for name, child in list(m._modules.items()):
                    if type(child) is nn.CrossEntropyLoss and child.weight is None and child.reduction == "mean" and _quiet(child):
                        m._modules[name] = _MaskedCE(child)
                        undo.append(("submodule", m, name, child))
                        counts["criterion"] += 1
    if bns:
        for m in cyl:
            counts["cylinder"] += _fuse_cylinder_block(m, undo)
    if bns and glue:
        counts["glue"] = _fuse_glue(model)
    if bns and forward and glue:   # the fused forwards are written against this package's glue helpers (see _glue_is_ours)
        for m in mods:
            counts["forward"] += _fuse_model_forward(m, undo)
    handle = (model.register_forward_pre_hook(_bump), model.register_forward_hook(_unbump)) if bns else None
    model.__dict__["_pcs_fused"] = {"counts": counts, "undo": undo, "hook": handle, "bns": bns}
    if bns and os.environ.get("PCS_FUSE_QUIET", "0") != "1" and not _REPORTED.get(type(model)):
        _REPORTED[type(model)] = True   # once per model class and process
        print("openpcseg_amd.fuse(%s): %s" % (type(model).__name__, ", ".join("%s %d" % kv for kv in counts.items())), file=sys.stderr)
    return dict(counts)


_REPORTED = {}


def unfuse(model):
    st = model.__dict__.pop("_pcs_fused", None)
    if st is None:
        return
    for u in reversed(st["undo"]):
        if u[0] == "class":
            u[1].__class__ = u[2]
            u[1].__dict__.pop("_pcs_plan", None)
            u[1].__dict__.pop("_pcs_orig_class", None)
        elif u[0] == "attr":
            setattr(u[1], u[2], u[3])
        else:
            u[1]._modules[u[2]] = u[3]
    if st["hook"] is not None:
        for h in st["hook"]:
            h.remove()
    for b in st["bns"]:
        if b.__dict__.pop("_pcs_bumped", False):
            b.num_batches_tracked.sub_(1)
    for m in model.modules():
        m.__dict__.pop("_pcs_fuse_seen", None)
        if isinstance(m, spnn.Conv3d):
            m.emit_bn_stats = False


# ---------------------------------------------------------------------------------------------------------------------
_AUTO = {"handle": None}


def install_auto_fuse():
    """`install_as_torchsparse(fuse=True)`: a process-wide forward pre-hook fuses every model the first time it is called (the
    first module called in a forward pass is the root). Costs one dict lookup per module call afterwards."""
    if _AUTO["handle"] is not None:
        return

    def hook(module, args):
        if "_pcs_fuse_seen" not in module.__dict__:
            fuse(module)
    _AUTO["handle"] = torch.nn.modules.module.register_module_forward_pre_hook(hook)


def uninstall_auto_fuse():
    if _AUTO["handle"] is not None:
        _AUTO["handle"].remove()
        _AUTO["handle"] = None
