"""Block fusion above the op boundary: BatchNorm + residual add + ReLU of the
reference's conv blocks as ONE forward apply pass and ONE backward apply pass over the voxel features.

`FusedBatchNorm` has the parameters / buffers / state_dict keys of nn.BatchNorm1d (and nn.SyncBatchNorm), so
reference checkpoints load; `sync=True` all-reduces the (sum, sum^2, count) vector over the default process group
between the statistics and the apply kernels -- SyncBatchNorm semantics, ONE small collective per layer and
direction on a dedicated process group, the global row count travelling inside the vector and staying on the device (R:pcseg/model/segmentor/voxel/minkunet/minkunet.py:23-25)."""
import torch
import torch.distributed as dist
from torch import nn
from torch.autograd import Function

from . import native
from .sparse import SparseTensor


def _world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def _syncing(sync):
    """True when a sync-mode layer must exchange its statistics. PCS_SYNC_WORLD1=1 (test rig): also in a one-rank process
    group, so that the whole collective path -- dedicated communicator, all-reduce, device-resident count -- runs over
    RCCL on a box with a single GPU."""
    if not sync or not (dist.is_available() and dist.is_initialized()):
        return False
    import os
    return dist.get_world_size() > 1 or os.environ.get("PCS_SYNC_WORLD1") == "1"


import os as _os
# BatchNorm backward statistics from the consumer's dgrad write-back (BNLink) [r6]. Built, parity-green, and SLOWER on the MI355X:
# bf16 step 52.3 -> 55.2 ms, fp32 115.4 -> 116.8 ms (profiles/round6_bn_link_ab.txt) -- the extra reads (the BatchNorm's input rows,
# its gate bits) and the per-tile reduction sit in the exposed tail of every dgrad tile, the pass they replace streams at 60-75 % of
# HBM peak. Off unless PCS_BN_BWD_LINK=1.
LINK_BN_BWD = _os.environ.get("PCS_BN_BWD_LINK", "0") == "1"

_STATS_GROUP = {}


def _stats_group():
    """A process group of its own for the BatchNorm statistics. On the default group the tiny (<= 6 KB) statistics
    all-reduces of backward queue FIFO behind DDP's 25 MB gradient buckets on the same RCCL communicator and stream
    -- each of the 63 BN layers then waits for a bucket to cross the xGMI ring before it can normalise its gradient.
    A second communicator (high-priority stream on RCCL) lets them overtake. Two communicators issued concurrently
    rely on every rank issuing its collectives in the same order: graphs that differ per rank (find_unused_parameters,
    data-dependent branches) should set PCS_BN_GROUP=0, the safe fallback, which keeps everything on the default group."""
    import os
    if os.environ.get("PCS_BN_GROUP", "1") == "0":
        return None
    # bound to the default group OBJECT (held in the cache entry and compared with `is`: after destroy_process_group() +
    # re-init CPython may hand the new group the old one's address, so an id() key could return a dead communicator)
    default = dist.distributed_c10d._get_default_group()
    key = (dist.get_world_size(), dist.get_backend())
    hit = _STATS_GROUP.get(key)
    g = hit[1] if hit is not None and hit[0] is default else None
    if g is None:
        _STATS_GROUP.clear()
        kw = {}
        if dist.get_backend() == "nccl":
            try:
                kw["pg_options"] = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
            except Exception:
                pass
        try:
            g = dist.new_group(backend=dist.get_backend(), **kw)  # collective: every rank reaches its first BN layer
        except Exception:  # an older / stricter torch.distributed: the statistics stay on the default group
            try:
                g = dist.new_group(backend=dist.get_backend())
            except Exception:
                g = False
        _STATS_GROUP[key] = (default, g)
    return g or None


class BNLink:
    """Hand-over between a fused BatchNorm and the ONE sparse convolution that consumes its output (conv -> BN -> ReLU -> conv,
    R:pcseg/model/segmentor/voxel/minkunet/minkunet.py:31-129) [r6]. Forward: the BatchNorm leaves what its backward statistics
    need (its input x, the ReLU gate bits, mean | invstd) here and hangs the link on its output tensor; a `_SparseConv` that takes
    that tensor counts itself as a consumer. Backward: when it is the only one, its dgrad launch -- which WRITES the BatchNorm's
    dy -- also leaves sum(g), sum(g xhat) per tile in its write-back (pcs_conv_gather_gemm_*_ex, bn_x), and the BatchNorm's
    backward reduces those instead of reading dy and x once more (pcs_bn_bwd_stats_*). The BatchNorm checks that the dy it is
    handed IS that launch's output, untouched (storage address + in-place version): any other consumer of the activation makes
    autograd sum gradients into another tensor (or in place), and the statistics pass runs as before."""
    __slots__ = ("x", "mask", "stat", "consumers", "partials", "dy_ptr", "dy_version")

    def __init__(self):
        self.x = self.mask = self.stat = self.partials = None
        self.consumers, self.dy_ptr, self.dy_version = 0, None, None


class _FusedBN(Function):
    """Feature tensors may be fp32, bf16 or fp16 (mixed precision: the half convolutions hand on halfs); statistics,
    scale / shift and running stats are fp32 / double whatever the storage format."""

    @staticmethod
    def forward(ctx, x, res, weight, bias, running_mean, running_var, eps, momentum, relu, sync, cache, level, pre=None,
                tail=None, link=None, in_slope=None):
        """in_slope: x is the output of a LeakyReLU that the producing convolution applied in its write-back; the gradient this
        node returns for x is then the gradient of the PRE-activation (the derivative rides in the backward apply pass), which is
        what that convolution's backward expects (functional._SparseConv, act_slope)."""
        """tail (n, ct): concat fusion -- the output is cat([bn(x), tail], 1), the BN result written straight into the left
        columns and `tail` copied to the right ones by the apply launch (no torch.cat pass); backward reads its dy out of
        the gradient of that buffer through a row stride and hands the right columns on as tail's gradient."""
        be = native.backend()
        x = x.contiguous()
        res = res.contiguous() if res is not None else None
        n, c = x.shape
        # [sum x | sum x^2 | n]: handed over by the producing convolution (its write-back computed them), else one pass
        syncing = _syncing(sync)
        count, count_dev, stat = float(n), None, None
        raw = pre is not None and pre.numel() != 2 * c + 1 and pre.numel() > 0 and pre.numel() % (2 * c) == 0
        if raw and not syncing and n > 0:
            # the producing convolution's per-tile partials: reduction and finalize in one launch (nothing to all-reduce)
            stat = be.bn_reduce_finalize(pre, c, n, eps, momentum, running_mean, running_var)
        elif raw:
            sums = be.bn_reduce_partials(pre, c, n)
        elif pre is not None and pre.numel() == 2 * c + 1:
            sums = pre.clone() if syncing else pre  # the all-reduce below works in place
        else:
            sums = be.bn_stats(x)
        if syncing:
            # ONE collective per layer and direction: the row count rides in the statistics vector and the global
            # count stays on the device (finalize / bwd_apply read it there) -- no count all-reduce, no host sync
            dist.all_reduce(sums, group=_stats_group())
            count_dev = sums[2 * c:]
        if stat is None:
            stat = be.bn_finalize(sums, count, eps, momentum, running_mean, running_var, count_dev=count_dev)
        # c % 32 == 0: the backward passes read the ReLU gate as a bit mask (1/32 of a tensor) instead of y
        if relu and c % 32 == 0 and c % 4 == 0:
            y, gate = be.bn_apply(x, res, stat, weight, bias, relu, want_mask=True, tail=tail)
        else:
            y = be.bn_apply(x, res, stat, weight, bias, relu, tail=tail)
            gate = (y if tail is None else y[:, :c].contiguous()) if relu else None
        ctx.save_for_backward(x, gate, stat, weight, count_dev)
        ctx.cfg = (count, relu, sync, res is not None, tail is not None)
        ctx.in_slope = in_slope
        ctx.link = None
        if link is not None and tail is None and n > 0 and c % 4 == 0 and hasattr(be, "bn_bwd_reduce_partials") and (
                not relu or (gate is not None and gate.dtype != y.dtype)):   # the gate as a bit mask (c % 32 == 0), or no ReLU
            link.x, link.mask, link.stat = x, (gate if relu else None), stat
            ctx.link = link
        return y

    @staticmethod
    def backward(ctx, dy):
        be = native.backend()
        x, gate, stat, weight, count_dev = ctx.saved_tensors
        count, relu, sync, has_res, has_tail = ctx.cfg
        c = x.shape[1]
        dtail = None
        if has_tail:
            dtail = dy[:, c:]   # the skip tensor's gradient: a view, summed into its other gradients by autograd
            dy = dy[:, :c]      # read in place through the row stride
        else:
            dy = dy.contiguous()
        link, local = ctx.link, None
        if link is not None:
            if (link.partials is not None and dy.data_ptr() == link.dy_ptr and dy._version == link.dy_version and
                    dy.shape == x.shape and dy.dtype == x.dtype):
                local = be.bn_bwd_reduce_partials(link.partials, c)   # left by the dgrad launch that wrote this dy
            link.partials = link.x = link.mask = link.stat = None
        if local is None:
            local = be.bn_bwd_stats(dy, x, gate, stat, relu)
        sums2 = local
        if _syncing(sync):
            sums2 = local.clone()
            dist.all_reduce(sums2, group=_stats_group())
        if ctx.in_slope is not None:
            dx, dres = be.bn_bwd_apply(dy, x, gate, stat, sums2, count, weight, relu, has_res, count_dev=count_dev, in_slope=ctx.in_slope)
        else:
            dx, dres = be.bn_bwd_apply(dy, x, gate, stat, sums2, count, weight, relu, has_res, count_dev=count_dev)
        dw = db = None
        if weight is not None:  # local sums: DDP averages parameter grads
            lw = getattr(local, "_pcs_f32", None)   # the HIP reduction leaves them in fp32 as well
            if lw is None or lw.dtype != weight.dtype:
                lw = local.to(weight.dtype)         # one cast for both halves
            dw, db = lw[c:], lw[:c]
        return dx, dres, dw, db, None, None, None, None, None, None, None, None, None, dtail, None, None


class FusedBatchNorm(nn.Module):
    """BatchNorm over SparseTensor features with optional fused residual add and ReLU:
    `bn(x)`, `bn(x, relu=True)`, `bn(x, residual=r, relu=True)`."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1, sync=False):
        super().__init__()
        self.num_features, self.eps, self.momentum, self.sync = num_features, eps, momentum, sync
        self.weight = nn.Parameter(torch.ones(num_features))
        self.bias = nn.Parameter(torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
        self.counted_by_parent = False

    def extra_repr(self):
        return "%d, eps=%g, momentum=%g, sync=%s" % (self.num_features, self.eps, self.momentum, self.sync)

    def forward(self, input, residual=None, relu=False, cat_with=None):
        """cat_with: a SparseTensor / tensor on the same coordinates; the result then carries cat([bn(x), cat_with], 1)
        (torchsparse.cat of the reference's decoder, fused into the apply pass)."""
        x = input.feats
        r = residual.feats if isinstance(residual, SparseTensor) else residual
        tail = cat_with.feats if isinstance(cat_with, SparseTensor) else cat_with
        if tail is not None and (x.shape[1] % 4 or tail.shape[1] % 4 or tail.shape[0] != x.shape[0]):
            # the concat-fused apply pass moves 16-byte pieces: other widths (the cr 1.6 configs: 409 + 204 ...) concatenate with torch
            y = self.forward(input, residual=residual, relu=relu)
            return input._like(torch.cat([y.feats, tail.to(y.feats.dtype)], dim=1))
        if self.training:
            if not self.counted_by_parent:  # a model may bump all its counters with one _foreach_add_ per step
                self.num_batches_tracked += 1
            # statistics handed over by the producing convolution are used only for the very tensor they describe
            # (same object, untouched since): `x.F = dropout(x.F)`, `x.feats += b` in between fall back to the stats pass
            pre = getattr(input, "bn_sums", None)
            if pre is not None:
                sums, of, ver = pre
                pre = sums if (of is x and x._version == ver) else None
            link = BNLink() if (LINK_BN_BWD and tail is None and torch.is_grad_enabled()) else None
            y = _FusedBN.apply(x, r, self.weight, self.bias, self.running_mean, self.running_var, self.eps,
                               self.momentum, relu, self.sync, input.cmaps, input.stride, pre, tail, link)
            if link is not None and link.x is not None:
                y._pcs_bn_link = link   # read by the sparse convolution that consumes y (functional._SparseConv)
        else:
            inv = torch.rsqrt(self.running_var.double() + self.eps)
            stat = torch.cat([self.running_mean.double(), inv]).contiguous()
            y = native.backend().bn_apply(x.contiguous(), r.contiguous() if r is not None else None, stat,
                                          self.weight, self.bias, relu, tail=tail.contiguous() if tail is not None else None)
        return input._like(y)


class _SkinnyLinear(Function):
    """y = x @ W^T + b for a tall-skinny problem (1.2 M rows, 480 -> 20 classes): hipBLASLt picks 32x32x256 /
    256x256x16 macro-tiles for it (forward 1.05 ms, dgrad 1.82 ms at 12-21 TFLOP/s). The fused conv kernels treat it
    as a K = 1 convolution over the identity map: forward + dgrad on the gather-GEMM, dW on the split-reduction wgrad."""

    @staticmethod
    def forward(ctx, x, weight, bias, cache, hd=None):
        """hd: the half dtype under autocast (the 16-bit MFMA kernel serves the forward when its shape rules allow)."""
        from .functional import _identity_map
        be = native.backend()
        km = _identity_map(x.shape[0], x.device, cache)
        w1 = weight.detach().float().t().contiguous().unsqueeze(0)  # (1, in, out)
        cin, cout = w1.shape[1], w1.shape[2]
        if hd is not None and be.conv_h_applies(cin, cout, 1):
            x = x.contiguous().to(hd)
            y = be.conv_gather_gemm_h(x, be.prepare_weights_h(w1, hd, transpose=False), 1, cout, km,
                                      bias.float() if bias is not None else None)
        else:
            x = x.contiguous().float()
            y = be.conv_gather_gemm(x, w1, km, bias.float() if bias is not None else None)
            if hd is not None:
                y = y.to(hd)
        ctx.save_for_backward(x, weight)
        ctx.km, ctx.hd, ctx.in_dtype = km, hd, x.dtype
        return y

    @staticmethod
    def backward(ctx, dy):
        be = native.backend()
        x, weight = ctx.saved_tensors
        dy = dy.contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = be.conv_gather_gemm(dy.float(), weight.detach().float().contiguous().unsqueeze(0), ctx.km)
        if ctx.needs_input_grad[1]:
            if x.dtype != torch.float32 and x.shape[1] % 4 == 0 and dy.shape[1] % 4 == 0:
                dw = be.conv_wgrad_h(x, dy.to(x.dtype), ctx.km, 0)[0].t()
            else:
                dw = be.conv_wgrad(x.float(), dy.float(), ctx.km, 0)[0].t()
            dw = dw.to(weight.dtype)
        if ctx.needs_input_grad[2]:
            db = dy.float().sum(0)
        return dx, dw, db, None, None


class _SkinnyLinearParts(Function):
    """y = cat(parts, 1) @ W^T + b without the concatenation: one gather-GEMM per column block of W, summed -- the
    classifier over [z1 | z2 | z3] (R:pcseg/model/segmentor/voxel/minkunet/minkunet.py:415-417) never materialises the
    (N, 480) tensor (2.8 GB written and read back per step at 1.4 M points) nor, in backward, the contiguous copies of
    its sliced gradient: every d(part) is written once, by its own launch."""

    @staticmethod
    def forward(ctx, weight, bias, cache, hd, *parts):
        from .functional import _identity_map
        be = native.backend()
        km = _identity_map(parts[0].shape[0], parts[0].device, cache)
        wt = weight.detach().float().t().contiguous()  # (in, out)
        saved, y, col = [], None, 0
        for i, x in enumerate(parts):
            cin = x.shape[1]
            w1 = wt[col:col + cin].unsqueeze(0)  # (1, cin, out): a contiguous row block
            b = bias.float() if (bias is not None and i == 0) else None
            # a part that ARRIVES in half runs on the 16-bit kernel; an fp32 part stays fp32 even under autocast: the
            # pass is HBM-bound, and casting first (read 4 + write 2 + read 2 bytes per element) costs twice the fp32 read
            if x.dtype != torch.float32 and be.conv_h_applies(cin, w1.shape[2], 1):
                x = x.contiguous()
                t = be.conv_gather_gemm_h(x, be.prepare_weights_h(w1.contiguous(), x.dtype, transpose=False), 1, w1.shape[2], km, b).float()
            else:
                x = x.contiguous().float()
                t = be.conv_gather_gemm(x, w1.contiguous(), km, b)
            y = t if y is None else y.add_(t)
            saved.append(x)
            col += cin
        ctx.save_for_backward(weight, *saved)
        ctx.km, ctx.hd, ctx.has_bias = km, hd, bias is not None
        return y.to(hd) if hd is not None else y

    @staticmethod
    def backward(ctx, dy):
        be = native.backend()
        weight, parts = ctx.saved_tensors[0], ctx.saved_tensors[1:]
        dy32 = dy.contiguous().float()
        w = weight.detach().float()  # (out, in)
        grads, dws, col = [], [], 0
        for i, x in enumerate(parts):
            cin = x.shape[1]
            dx = None
            if ctx.needs_input_grad[4 + i]:
                dx = be.conv_gather_gemm(dy32, w[:, col:col + cin].contiguous().unsqueeze(0), ctx.km)
            grads.append(dx)
            if ctx.needs_input_grad[0]:
                if x.dtype != torch.float32 and cin % 4 == 0 and dy.shape[1] % 4 == 0:
                    dws.append(be.conv_wgrad_h(x, dy32.to(x.dtype), ctx.km, 0)[0].t())
                else:
                    dws.append(be.conv_wgrad(x.float(), dy32, ctx.km, 0)[0].t())
            col += cin
        dw = torch.cat(dws, dim=1).to(weight.dtype) if ctx.needs_input_grad[0] else None
        db = dy32.sum(0) if (ctx.has_bias and ctx.needs_input_grad[1]) else None
        return (dw, db, None, None) + tuple(grads)


def devoxelized_linear(weight, col, xf, idx, wts, cache):
    """devoxelize(xf) @ W_i^T computed as devoxelize(xf @ W_i^T), W_i = weight[:, col : col + C_i] of an nn.Linear weight:
    trilinear devoxelisation is linear over the voxel features (fixed per-point weights), so it commutes with the classifier --
    the class scores are formed on the VOXELS (36 k / 329 k / 1.16 M rows of 256 / 128 / 96 channels -> num_class) and only
    num_class channels per point are interpolated, instead of interpolating 480 channels per point and contracting them there
    (2.7 GB of point features written, read by the classifier, and the same again as gradients in backward). Same function; the
    fp32 rounding order differs. xf (V, C_i) voxel features, idx / wts (N, 8) the points' corner map. The bias is added on the
    points by the caller: the weights of a point with missing corners do not sum to one. cache: dict that keeps the identity map
    of this row count (pass the tensor's per-forward `kmaps`: built once per level and step, freed with the step)."""
    from . import functional as F_
    cin, cout = xf.shape[1], weight.shape[0]
    w = weight[:, col:col + cin]
    ok = (xf.is_cuda and xf.dim() == 2 and xf.dtype in (torch.float32, torch.bfloat16, torch.float16) and
          xf.shape[0] >= 4096 and cin % 4 == 0 and cout % 4 == 0)
    if ok:
        v = _SkinnyLinear.apply(xf, w, None, cache, xf.dtype if xf.dtype != torch.float32 else None)
    else:
        v = torch.nn.functional.linear(xf.float(), w.float())
    return F_.spdevoxelize(v.float(), idx, wts)


class FusedLinear(nn.Linear):
    """nn.Linear (same parameters / state_dict keys) whose fp32 device path runs on the fused conv kernels when the
    row count dwarfs the feature sizes and the shapes are 16-byte granular; anything else is nn.Linear."""

    def __init__(self, in_features, out_features, bias=True):
        super().__init__(in_features, out_features, bias)
        self._maps = {}  # identity maps by row count (a handful of distinct batch sizes per run)

    def forward(self, x):
        ok = (x.is_cuda and x.dtype in (torch.float32, torch.bfloat16, torch.float16) and x.dim() == 2 and
              x.shape[0] >= 4096 and self.in_features % 4 == 0 and self.out_features % 4 == 0)
        if not ok:
            return super().forward(x)
        if len(self._maps) > 8:
            self._maps.clear()
        hd = None
        if x.dtype != torch.float32:
            hd = x.dtype
        elif torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") in (torch.bfloat16, torch.float16):
            hd = torch.get_autocast_dtype("cuda")
        return _SkinnyLinear.apply(x, self.weight, self.bias, self._maps, hd)

    def devoxelized_part(self, col, xf, idx, wts, cache=None):
        """One term of forward(cat([devoxelize(x_i) for i], 1)) = sum_i devoxelize(x_i @ W_i^T) + b, W_i = the column
        block [col, col + C_i) of the weight (`devoxelized_linear`). cache: a per-forward dict for the identity map of this row
        count (the tensor's `kmaps`); default: the module's own small cache."""
        if cache is None:
            if len(self._maps) > 8:
                self._maps.clear()
            cache = self._maps
        return devoxelized_linear(self.weight, col, xf, idx, wts, cache)

    def sum_devoxelized(self, terms):
        y = terms[0]
        for t in terms[1:]:
            y = y + t
        return y + self.bias if self.bias is not None else y

    def forward_parts(self, parts):
        """forward(torch.cat(parts, 1)) without building the concatenation (column blocks of the weight)."""
        parts = list(parts)
        ok = (all(x.is_cuda and x.dim() == 2 and x.dtype in (torch.float32, torch.bfloat16, torch.float16) and
                  x.shape[1] % 4 == 0 for x in parts) and parts[0].shape[0] >= 4096 and self.out_features % 4 == 0 and
              sum(x.shape[1] for x in parts) == self.in_features)
        if not ok:
            return self.forward(torch.cat(parts, dim=1))
        if len(self._maps) > 8:
            self._maps.clear()
        hd = None
        if any(x.dtype != torch.float32 for x in parts):
            hd = next(x.dtype for x in parts if x.dtype != torch.float32)
        elif torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") in (torch.bfloat16, torch.float16):
            hd = torch.get_autocast_dtype("cuda")
        return _SkinnyLinearParts.apply(self.weight, self.bias, self._maps, hd, *parts)
