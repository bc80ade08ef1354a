"""Functional operator API of the reference (`torchsparse.nn.functional`), on the HIP backend.

Same names, argument meaning and error behaviour as
  sphash          TS:torchsparse/nn/functional/hash.py:10-37
  sphashquery     TS:torchsparse/nn/functional/query.py:8-33
  spcount         TS:torchsparse/nn/functional/count.py:8-16
  spvoxelize      TS:torchsparse/nn/functional/voxelize.py:10-56
  spdevoxelize / calc_ti_weights   TS:torchsparse/nn/functional/devoxelize.py:10-98
  spdownsample    TS:torchsparse/nn/functional/downsample.py:11-52
  conv3d          TS:torchsparse/nn/functional/conv.py:16-205
Internals differ: every op is one or a few C-ABI calls (openpcseg_amd.native); the rulebook is
built by a fused probe/compaction pass instead of kernel_hash -> hashquery -> sum -> nonzero,
and the convolution is an output-stationary fused gather-GEMM (no per-offset launches, no
`nbsizes.cpu()` per call).
Autocast: the reference casts op inputs to fp16 under AMP (`custom_fwd(cast_inputs=torch.half)`). Here conv3d follows
the autocast dtype (bf16 or fp16) on the 16-bit MFMA kernels (`_SparseConv`); voxelize / devoxelize keep fp32 (at
least the reference's precision).
"""
import os

import torch
from torch.autograd import Function
from torch.amp import custom_bwd, custom_fwd

from . import native
from .sparse import SparseTensor, get_kernel_offsets, make_ntuple

__all__ = ["sphash", "sphashquery", "spcount", "spvoxelize", "spdevoxelize", "calc_ti_weights",
           "spdownsample", "prebuild_coords", "conv3d", "relu", "leaky_relu"]


def _be():
    return native.backend()


# ---------------------------------------------------------------------------------------------
def sphash(coords, offsets=None):
    assert coords.dtype == torch.int, coords.dtype
    assert coords.ndim == 2 and coords.shape[1] == 4, coords.shape
    coords = coords.contiguous()
    if offsets is None:
        return _be().hash(coords)
    assert offsets.dtype == torch.int, offsets.dtype
    assert offsets.ndim == 2 and offsets.shape[1] == 3, offsets.shape
    return _be().kernel_hash(coords, offsets.contiguous())


def sphashquery(queries, references):
    queries = queries.contiguous()
    references = references.contiguous()
    sizes = queries.size()
    return _be().hash_query(queries.view(-1), references).view(*sizes)


def spcount(coords, num):
    return _be().count(coords.contiguous(), num)


# ---------------------------------------------------------------------------------------------
class _Voxelize(Function):
    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, feats, coords, counts):
        feats = feats.contiguous()
        holder = coords  # the caller's tensor outlives this call (idx_query cache of point_to_voxel)
        coords = coords.contiguous().int()
        out = _be().voxelize_fwd(feats, coords, counts, cache_on=holder)
        ctx.for_backwards = (coords, counts, feats.shape[0])
        return out

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, grad_output):
        coords, counts, n = ctx.for_backwards
        return _be().voxelize_bwd(grad_output.contiguous(), coords, counts, n), None, None


def spvoxelize(feats, coords, counts):
    return _Voxelize.apply(feats, coords, counts)


class _Devoxelize(Function):
    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, feats, coords, weights):
        feats = feats.contiguous()
        coords = coords.contiguous().int()
        weights = weights.contiguous()
        out = _be().devoxelize_fwd(feats, coords, weights)
        ctx.for_backwards = (coords, weights, feats.shape[0])
        return out

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, grad_output):
        coords, weights, m = ctx.for_backwards
        return _be().devoxelize_bwd(grad_output.contiguous(), coords, weights, m), None, None


def spdevoxelize(feats, coords, weights):
    return _Devoxelize.apply(feats, coords, weights)


def calc_ti_weights(coords, idx_query, scale=1):
    """(8,N) trilinear weights; coords (N,>=3) float, idx_query (8,N) with -1 = missing corner."""
    with torch.no_grad():
        return _be().ti_weights(coords.float(), idx_query.long(), scale)


# ---------------------------------------------------------------------------------------------
def spdownsample(coords, stride=2, kernel_size=2, tensor_stride=1):
    stride = make_ntuple(stride, ndim=3)
    kernel_size = make_ntuple(kernel_size, ndim=3)
    tensor_stride = make_ntuple(tensor_stride, ndim=3)
    sample_stride = [stride[k] * tensor_stride[k] for k in range(3)]
    if all(stride[k] in [1, kernel_size[k]] for k in range(3)):
        return _be().downsample(coords, sample_stride)
    offsets = get_kernel_offsets(kernel_size, tensor_stride, device=coords.device)
    coords_min = torch.min(coords[:, :3], dim=0).values.int()
    return _be().downsample(coords, sample_stride, offsets, coords_min)


# ---------------------------------------------------------------------------------------------
class KmapEntry(list):
    """kmaps[(stride, kernel_size, conv_stride, dilation)] = [nbmaps, nbsizes, (n_in, n_out)]
    like the reference (conv.py:174-176), plus the native maps the kernels use:
      .fwd  pairs (in_row, out_row), out ascending within an offset  (= nbmaps)
      .rev  pairs (out_row, in_row), in ascending within an offset   (built on first use;
            needed by dgrad and by transposed convolutions)"""

    def __init__(self, fwd, in_coords, out_coords, offsets, symmetric=False, hint_key=None):
        # item 0 (nbmaps) is read through __getitem__: the native map hands out its exact-size pair list lazily
        super().__init__([None, fwd.nbsizes, (in_coords.shape[0], out_coords.shape[0])])
        self.fwd = fwd
        self._rev = None
        self._ctx = (in_coords, out_coords, offsets)
        self._hint_key = hint_key
        # submanifold map (same coordinate tensor on both sides) with point-symmetric offsets: the input-sorted
        # map is a slice permutation of the forward map, no second probe pass
        self._mirror = symmetric and in_coords is out_coords

    def __getitem__(self, i):
        if isinstance(i, int) and i in (0, -3):
            return self.fwd.pairs
        return super().__getitem__(i)

    def __iter__(self):
        return iter([self[0], super().__getitem__(1), super().__getitem__(2)])

    @property
    def rev(self):
        if self._rev is None:
            in_coords, out_coords, offsets = self._ctx
            if self._mirror:
                self._rev = self.fwd.mirror()
            else:
                key = None if self._hint_key is None else self._hint_key + ("rev",)
                self._rev = _be().build_kmap(out_coords, in_coords, -offsets, hint_key=key)
        return self._rev


def build_kernel_map(in_coords, out_coords, kernel_size, in_stride, dilation):
    offsets = get_kernel_offsets(kernel_size, stride=in_stride, dilation=dilation, device="cpu")
    symmetric = bool(torch.equal(offsets.flip(0), -offsets))  # odd kernel sizes
    offsets = (offsets.pin_memory() if in_coords.is_cuda else offsets).to(in_coords.device, non_blocking=True)
    hint_key = (tuple(kernel_size), tuple(in_stride), tuple(dilation), in_coords is out_coords)  # map family
    fwd = _be().build_kmap(in_coords, out_coords, offsets, hint_key=hint_key, symmetric=symmetric and in_coords is out_coords)
    return KmapEntry(fwd, in_coords, out_coords, offsets, symmetric, hint_key)


_HALF = (torch.bfloat16, torch.float16)

# fp32 weight gradient of conv3d. "fp32" (library default): fp32 MFMA arithmetic (wgrad2_kernel). "bf16x3": layers with
# >= 96 input and output channels take pcs_conv_wgrad_f32_bf16x3 -- every fp32 operand split into three bf16 planes, six plane
# products accumulated in fp32 on the 16-bit MFMAs: fp32-grade (error vs float64 at most twice the fp32 MFMA path's,
# tests/test_dense_parity.py::test_conv_backward_dense_map), 1.3-1.4x faster there; thin layers stay on wgrad2 (slower
# on the split path). Opt-in only: a caller that selects it states so (bench.py's config.wgrad).
_WGRAD_POLICY = {"mode": "fp32"}


def set_wgrad_policy(mode):
    if mode not in ("fp32", "bf16x3"):
        raise ValueError("wgrad policy must be 'fp32' or 'bf16x3'")
    _WGRAD_POLICY["mode"] = mode


def get_wgrad_policy():
    return _WGRAD_POLICY["mode"]


# fp32 forward / input-gradient convolution. "fp32" (library default): fp32 MFMA arithmetic (conv_os5_kernel). "bf16x3": layers
# the split kernel serves (cin % 8 == 0, cin >= 32, cout >= 32) run conv_os5x_kernel -- fp32 in and out, every operand as three
# bf16 planes, six plane products accumulated in fp32 on the 16-bit MFMAs (csrc/conv_wave5x.hip): fp32-grade, not bit-identical
# to an fp32 FMA chain. Opt-in only; a caller that selects it states so (bench.py's `fp32_bf16x3` record).
_CONV_POLICY = {"mode": "fp32"}


def set_conv_policy(mode):
    if mode not in ("fp32", "bf16x3"):
        raise ValueError("conv policy must be 'fp32' or 'bf16x3'")
    _CONV_POLICY["mode"] = mode


def get_conv_policy():
    return _CONV_POLICY["mode"]


def _wgrad_split(cin, cout):
    return _WGRAD_POLICY["mode"] == "bf16x3" and cin >= 96 and cout >= 96 and cin % 4 == 0 and cout % 4 == 0


def _amp_dtype(t):
    """The half dtype this op computes in, or None for fp32: the input's own dtype when it already is half, else the
    autocast dtype while autocast is on (the reference casts op inputs to half under AMP,
    TS:torchsparse/nn/functional/conv.py:19 `custom_fwd(cast_inputs=torch.half)`)."""
    if t.dtype in _HALF:
        return t.dtype
    if t.is_cuda and torch.is_autocast_enabled("cuda"):
        d = torch.get_autocast_dtype("cuda")
        return d if d in _HALF else None
    return None


class _WeightPrep:
    """Prepared copies of the convolution weights (dgrad's per-offset transposes in fp32, the fragment-ordered half weights of
    forward and dgrad under autocast), refreshed for ALL layers by one launch (`native.weights_multi`) instead of inside every
    layer call like the reference (TS:torchsparse/backend/convolution/convolution_cuda.cu:196-206,
    TS:torchsparse/nn/functional/conv.py:19).

    When a copy is stale: the copies belong to one PASS over the model. A pass ends the moment a (weight, copy) pair that was
    already handed out in it is asked for again -- the next forward has started -- and the first request of a new pass
    re-prepares every known copy from the live weights, whatever happened to them in between: an optimizer step, `load_state_dict`,
    but also writes the autograd version counter does not see (`w.data.copy_()`, `w.data = t`, EMA / SWA swaps,
    `vector_to_parameters`, multi-tensor optimizers writing through raw pointers). Inside a pass a copy is additionally refreshed
    when the parameter's version counter or storage address changed. A training step pays the one launch it always paid
    (the optimizer made everything stale anyway); inference pays one ~0.1 ms launch per forward. `invalidate()` forces a refresh.
    What the pass rule does NOT see: a SECOND model written through `.data` after the first model's pass began and run before any
    copy of the first is asked for twice (an EMA / teacher copy updated by hand and evaluated in the same iteration). Every
    `torch.optim.Optimizer.step` ends the pass through a global post-step hook (registered below), which covers the usual EMA
    order (student step, then teacher update, then teacher forward); a hand-written update with no optimizer in between must
    call `openpcseg_amd.functional.invalidate_prepared_weights()`.
    Only leaf fp32 (K, A, B) device parameters are cached; anything else (padded copies, 2-D weights, other backends) takes the
    per-call path. PCS_WEIGHT_PREP=0 switches the cache off (A/B)."""

    def __init__(self):
        self.entries = {}   # id(weight) -> [weakref, (pass, version, data_ptr) the copies were made at, {key: tensor}, keys handed out in that pass]
        self.pass_id = 0

    @staticmethod
    def usable(be, weight):
        return (hasattr(be, "weights_multi") and isinstance(weight, torch.nn.Parameter) and weight.is_cuda and weight.dim() == 3 and
                weight.dtype == torch.float32 and weight.is_contiguous() and os.environ.get("PCS_WEIGHT_PREP", "1") != "0")

    def invalidate(self):
        """Every prepared copy is stale from now on (call after writing weights behind autograd's back mid-pass)."""
        self.pass_id += 1

    def _stamp(self, w):
        return (self.pass_id, w._version, w.data_ptr())

    def get(self, be, weight, key):
        """key: ("t",) or (half dtype, transpose)."""
        import weakref
        e = self.entries.get(id(weight))
        if e is None or e[0]() is not weight:
            # a new parameter: drop the copies of parameters that no longer exist (models come and go in one process)
            self.entries = {i: v for i, v in self.entries.items() if v[0]() is not None}
            e = [weakref.ref(weight), None, {}, set()]
            self.entries[id(weight)] = e
        if e[1] is not None and e[1][0] == self.pass_id and key in e[3]:
            self.pass_id += 1   # this copy was already handed out in the current pass: a new pass over the model has begun
        if e[1] == self._stamp(weight) and key in e[2]:
            e[3].add(key)
            return e[2][key]
        if key not in e[2]:
            e[2][key] = be.prepared_weights_buffer(weight, *((("t", False)) if key == ("t",) else key))
        jobs, marks = [], []
        for v in self.entries.values():
            w = v[0]()
            if w is None or w.device != weight.device:
                continue
            stamp = self._stamp(w)
            if v[1] != stamp or (v is e):
                for k_, dst in v[2].items():
                    if v[1] != stamp or k_ == key:
                        jobs.append((w.detach(), dst, "t" if k_ == ("t",) else k_[0], False if k_ == ("t",) else k_[1]))
                marks.append((v, stamp))
        be.weights_multi(jobs)
        for v, stamp in marks:
            if v[1] != stamp:
                v[3] = set()
            v[1] = stamp
        e[3].add(key)
        return e[2][key]


_WEIGHT_PREP = _WeightPrep()


def invalidate_prepared_weights():
    """Every prepared weight copy is rebuilt at its next use (see _WeightPrep)."""
    _WEIGHT_PREP.invalidate()


try:   # an optimizer step ends the pass: whatever is run next re-prepares from the live weights
    from torch.optim.optimizer import register_optimizer_step_post_hook as _reg_post_step
    _reg_post_step(lambda opt, args, kwargs: _WEIGHT_PREP.invalidate())
except ImportError:   # older torch: the pass rule and the version counters alone
    pass


class _SparseConv(Function):
    """out = conv(input) over a kernel map; backward = dgrad (same fused kernel on the other
    map, per-offset transposed weights) + wgrad (split reduction).

    Mixed precision (autocast, or half features in): layers the half kernels serve (cin, cout >= 64-class shapes,
    pcs_conv_h_applies) run on the 16-bit MFMA kernels -- features and outputs in bf16 / fp16, weights re-packed
    from the fp32 master copy per call, fp32 accumulation, fp32 weight gradient; the remaining thin layers are
    computed in fp32 and their output rounded to the half dtype, like the reference's half pipeline would hand on."""

    @staticmethod
    def forward(ctx, input, weight, entry, transposed, want_stats=False, with_skip=False, act_slope=None):
        """want_stats: also return the BatchNorm statistics of the output when the kernel produced them in its write-back
        -- on the HIP backend the per-tile partials ([tiles][2][cout] float64; `_FusedBN` reduces them), on others the
        reduced vector [sum x | sum x^2 | n] --, else an empty tensor (the BatchNorm then runs its own pass).
        with_skip: also return the input itself (an alias) as the LAST output, for the caller's residual / skip path
        (`relu(net(x) + downsample(x))`, R:pcseg/model/segmentor/voxel/minkunet/minkunet.py:88-129). Both uses of x then hang on
        this one autograd node, which receives the skip path's gradient together with the convolution's and lets the dgrad
        kernel add it in its write-back (pcs_conv_gather_gemm_*_ex) -- instead of autograd summing two gradient tensors with an
        elementwise kernel per block.
        act_slope: the kernel applies LeakyReLU(act_slope) in its write-back (conv -> LeakyReLU -> BatchNorm1d,
        R:pcseg/model/segmentor/voxel/cylinder3d/cylinder_ts.py:88-190). CONTRACT: the gradient this node is handed is then the
        gradient of the PRE-activation -- the consumer (a fused BatchNorm with in_slope, fused._FusedBN) has multiplied the
        activation's derivative in. Only block_fusion's Cylinder forwards use it, with exactly that consumer; shapes whose kernel
        takes no write-back extras are refused by the caller (`conv_act_fusable`)."""
        be = _be()
        hd = _amp_dtype(input)
        w3 = weight if weight.dim() == 3 else weight.unsqueeze(0)
        k, cin, cout = w3.shape
        kmap = entry.rev if transposed else entry.fwd
        # the input is a fused BatchNorm's output: when this convolution is its only consumer, its dgrad write-back leaves that
        # BatchNorm's backward statistics (fused.BNLink)
        ctx.bn_link = getattr(input, "_pcs_bn_link", None)
        if ctx.bn_link is not None:
            ctx.bn_link.consumers += 1
        got = [] if want_stats else None
        kw = {"bn_sums": got} if want_stats else {}
        if want_stats and getattr(be, "supports_bn_raw", False):
            kw["bn_raw"] = True   # the per-tile partials themselves: the BatchNorm reduces and finalizes them in ONE launch
        if act_slope is not None:
            kw["act_slope"] = float(act_slope)
        if hd is not None and input.is_cuda and be.conv_h_applies(cin, cout, k):
            x = input.contiguous().to(hd)
            if _WeightPrep.usable(be, weight):
                wp = _WEIGHT_PREP.get(be, weight, (hd, False))
            else:
                wp = be.prepare_weights_h(w3.detach().float().contiguous(), hd, transpose=False)
            out = be.conv_gather_gemm_h(x, wp, k, cout, kmap, **kw)
        elif hd is None and input.is_cuda and _CONV_POLICY["mode"] == "bf16x3" and be.conv_x3_applies(cin, cout, k):
            x = input.contiguous().float()
            wp = be.prepare_weights_x3(w3.detach().float().contiguous(), transpose=False)
            out = be.conv_gather_gemm_x3(x, wp, k, cout, kmap, **kw)
        else:
            x = input.contiguous().float()
            out = be.conv_gather_gemm(x, w3.float().contiguous(), kmap, **kw)
            if hd is not None:
                out = out.to(hd)
                got = [] if want_stats else None  # statistics of the fp32 values, not of the rounded ones: not used
        ctx.for_backwards = (x, weight, entry, transposed, hd)
        ctx.with_skip = with_skip
        ctx.in_dtype = input.dtype
        ctx.set_materialize_grads(False)  # no zero-filled "gradient" for the statistics vector on every backward
        outs = [out]
        if want_stats:
            sums = got[0] if got else torch.empty(0, dtype=torch.float64, device=out.device)
            ctx.mark_non_differentiable(sums)
            outs.append(sums)
        if with_skip:
            outs.append(input.view_as(input))
        return outs[0] if len(outs) == 1 else tuple(outs)

    @staticmethod
    def backward(ctx, grad_output, *rest):
        grad_skip = rest[-1] if ctx.with_skip and rest else None
        if grad_output is None:
            return grad_skip, None, None, None, None, None, None
        be = _be()
        x, weight, entry, transposed, hd = ctx.for_backwards
        w3 = weight if weight.dim() == 3 else weight.unsqueeze(0)
        k, cin, cout = w3.shape
        grad_input = grad_weight = None
        need_dx, need_dw = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        # (round 3 tried the weight gradient on a second HIP stream beside dgrad to fill each other's launch tails; dropped:
        # a launch that shares the chip has no duration of its own for the roofline, and the lazily built per-map caches
        # would need cross-stream ordering)
        if need_dx:
            dmap = entry.fwd if transposed else entry.rev
            out_dtype = x.dtype if hd is None else hd   # the gradient leaves in the dtype the forward input arrived in
            rides = (grad_skip is not None and grad_skip.is_cuda and grad_skip.dtype == out_dtype and
                     hasattr(be, "conv_supports_addend"))   # the skip gradient as the dgrad kernel's write-back addend
            link, bnb, bnb_out = ctx.bn_link, None, []
            if (link is not None and link.consumers == 1 and link.x is not None and link.x.dtype == out_dtype and
                    tuple(link.x.shape) == (dmap.n_dst, cin) and hasattr(be, "conv_emits_stats") and
                    (grad_skip is None or rides)):
                bnb = (link.x, link.mask, link.stat)
            if hd is not None and be.conv_h_applies(cout, cin, k):
                if _WeightPrep.usable(be, weight):
                    wp = _WEIGHT_PREP.get(be, weight, (hd, True))
                else:
                    wp = be.prepare_weights_h(w3.detach().float().contiguous(), hd, transpose=True)
                ok = be.conv_supports_addend(cout, cin, k, 1) if (rides or bnb) else False
                if bnb is not None and not (ok and be.conv_emits_stats(cout, cin, k, dmap, hd)):
                    bnb = None
                kw = {}
                if rides and ok:
                    kw["addend"], grad_skip = grad_skip, None
                if bnb is not None and grad_skip is None:
                    kw["bn_bwd"], kw["bn_bwd_out"] = bnb, bnb_out
                grad_input = be.conv_gather_gemm_h(grad_output.contiguous().to(hd), wp, k, cin, dmap, **kw)
            elif hd is None and grad_output.is_cuda and _CONV_POLICY["mode"] == "bf16x3" and be.conv_x3_applies(cout, cin, k):
                wp = be.prepare_weights_x3(w3.detach().float().contiguous(), transpose=True)
                grad_input = be.conv_gather_gemm_x3(grad_output.contiguous().float(), wp, k, cin, dmap)
            else:
                if _WeightPrep.usable(be, weight):
                    wt = _WEIGHT_PREP.get(be, weight, ("t",))
                else:
                    wt = be.transpose_weights(w3.detach().float().contiguous())
                ok = (hd is None and out_dtype == torch.float32 and be.conv_supports_addend(cout, cin, k, 0)) if (rides or bnb) else False
                if bnb is not None and not (ok and be.conv_emits_stats(cout, cin, k, dmap, None)):
                    bnb = None
                kw = {}
                if rides and ok:
                    kw["addend"], grad_skip = grad_skip, None
                if bnb is not None and grad_skip is None:
                    kw["bn_bwd"], kw["bn_bwd_out"] = bnb, bnb_out
                grad_input = be.conv_gather_gemm(grad_output.contiguous().float(), wt, dmap, **kw)
            grad_input = grad_input.to(out_dtype)
            if bnb_out:   # this tensor IS the BatchNorm's dy: its backward recognises it by storage address and version
                link.partials, link.dy_ptr, link.dy_version = bnb_out[0], grad_input.data_ptr(), grad_input._version
            if grad_skip is not None:   # a kernel that takes no addend (generic shapes, the split kernels, other backends)
                grad_input = grad_input + grad_skip.to(out_dtype)
        elif grad_skip is not None:
            grad_input = grad_skip
        if need_dw:
            # fwd pairs are (in_row, out_row) of the NON-transposed conv; a transposed conv's
            # input lives on the out rows (column 1)
            a_col = 1 if transposed else 0
            if x.dtype in _HALF and cin % 4 == 0 and cout % 4 == 0:
                grad_weight = be.conv_wgrad_h(x, grad_output.contiguous().to(x.dtype), entry.fwd, a_col)
            else:
                grad_weight = be.conv_wgrad(x.float(), grad_output.contiguous().float(), entry.fwd, a_col,
                                            split=_wgrad_split(cin, cout))
            grad_weight = grad_weight.view_as(weight).to(weight.dtype)
        return grad_input, grad_weight, None, None, None, None, None


def conv_act_fusable(feats, weight):
    """The convolution of these operands runs on a kernel that takes the write-back extras (LeakyReLU, addend)."""
    be = _be()
    if not (feats.is_cuda and hasattr(be, "conv_supports_addend") and weight.dim() == 3):
        return False
    k, cin, cout = weight.shape
    if _channel_padding(feats, weight) != (0, 0) or (_amp_dtype(feats) is None and _CONV_POLICY["mode"] == "bf16x3"):
        return False
    hd = _amp_dtype(feats)
    if hd is not None and be.conv_h_applies(cin, cout, k):
        return True
    return be.conv_supports_addend(cin, cout, k, 0)


def _identity_map(n, device, cache):
    """K = 1 map (i, i): lets the split-reduction wgrad kernel compute x^T @ dy of a 1x1x1 convolution."""
    key = ("_pcs_identity", n, str(device))
    km = cache.get(key)
    if km is None:
        idx = torch.arange(n, dtype=torch.int32, device=device)
        km = native.KernelMap(torch.stack([idx, idx], dim=1).contiguous(),
                              native._h2d([0, n], torch.int32, device), [0, n],
                              native._h2d([n], torch.int64, device), n, n)
        cache[key] = km
    return km


class _PointwiseConv(Function):
    """1x1x1 convolution = feats @ weight (TS:torchsparse/nn/functional/conv.py:135-140). Forward and dgrad stay
    dense GEMMs (hipBLASLt through torch); the weight gradient x^T @ dy has a (Cin, Cout) output and a
    million-row contraction, which hipBLASLt runs on ~12 workgroups -- the split-reduction wgrad kernel is used."""

    @staticmethod
    def forward(ctx, feats, weight, cache):
        hd = _amp_dtype(feats)
        ctx.save_for_backward(feats, weight)
        ctx.cache, ctx.hd = cache, hd
        if hd is None:
            return feats.float().matmul(weight.float())
        return feats.to(hd).matmul(weight.to(hd))  # hipBLASLt, fp32 accumulate

    @staticmethod
    def backward(ctx, grad_output):
        feats, weight = ctx.saved_tensors
        grad_output = grad_output.contiguous()
        hd = ctx.hd
        gin = gw = None
        if ctx.needs_input_grad[0]:
            if hd is None:
                gin = grad_output.float().matmul(weight.float().t())
            else:
                gin = grad_output.to(hd).matmul(weight.to(hd).t())
        if ctx.needs_input_grad[1]:
            km = _identity_map(feats.shape[0], feats.device, ctx.cache)
            ca, cb = feats.shape[1], grad_output.shape[1]
            fa, gb = feats.contiguous(), grad_output
            pa, pb = (-ca) % 4, (-cb) % 4
            if (pa or pb) and feats.is_cuda and ca + pa >= 32:
                # widths that are not 16-byte granular (cr 1.6: 153, 409, 613 ...): zero columns bring the operands onto the MFMA
                # weight-gradient kernels (the generic kernel ran these 1x1x1 layers at 16-19 TFLOP/s); the padding's rows /
                # columns of the result are cut off again
                fa = torch.nn.functional.pad(fa, (0, pa)) if pa else fa
                gb = torch.nn.functional.pad(gb, (0, pb)) if pb else gb
            if hd is not None and fa.shape[1] % 4 == 0 and gb.shape[1] % 4 == 0:
                gw = _be().conv_wgrad_h(fa.to(hd), gb.to(hd), km, 0)[0]
            else:
                gw = _be().conv_wgrad(fa.float(), gb.float(), km, 0)[0]
            gw = gw[:ca, :cb].to(weight.dtype)
        return gin, gw, None


def _channel_padding(feats, weight):
    """(pad_in, pad_out) that bring a layer whose channel counts are not 16-byte granular onto the MFMA kernels, else (0, 0).
    The cr 1.6 model-zoo configs (R:tools/cfgs/voxel/waymo/minkunet_mk34_cr16.yaml:20, R:tools/cfgs/fusion/*/spvcnn_mk34_cr16.yaml:
    51 / 102 / 153 / 204 / 409 channels) would otherwise run every layer on the generic kernels (conv_os4_kernel / conv_block:
    5-25 TFLOP/s) in fp32 even under autocast. Zero channels change nothing: padded input columns meet zero weight rows, padded
    output columns are cut off again; autograd differentiates the pad and the slice (gradients of the padding are dropped)."""
    if not feats.is_cuda or weight.dim() != 3 or weight.shape[0] > 32:
        return 0, 0
    k, cin, cout = weight.shape
    q = 8 if _amp_dtype(feats) is not None else 4   # 16 bytes of a row
    pin, pout = (-cin) % q, (-cout) % q
    if (pin == 0 and pout == 0) or cin + pin < 32:
        return 0, 0
    return pin, pout


def _sparse_conv(feats, weight, entry, transposed, bn_stats, with_skip=False, act_slope=None):
    """-> (out, bn_sums or None[, skip alias of feats when with_skip])."""
    pin, pout = _channel_padding(feats, weight)
    if act_slope is not None:
        assert not (pin or pout), "act_slope: the caller checks conv_act_fusable()"
        outs = _SparseConv.apply(feats, weight, entry, transposed, bool(bn_stats), bool(with_skip), float(act_slope))
        outs = outs if isinstance(outs, tuple) else (outs,)
        sums = outs[1] if bn_stats else None
        res = (outs[0], (sums if sums is not None and sums.numel() else None))
        return res + (outs[-1],) if with_skip else res
    if pin or pout:
        cout = weight.shape[2]
        padded = torch.nn.functional.pad(feats, (0, pin)) if pin else feats
        weight = torch.nn.functional.pad(weight, (0, pout, 0, pin))
        out = _SparseConv.apply(padded, weight, entry, transposed)
        res = ((out[:, :cout].contiguous() if pout else out), None)   # the epilogue statistics would cover the padded columns: not used
        return res + (feats,) if with_skip else res
    if not bn_stats and not with_skip:
        return _SparseConv.apply(feats, weight, entry, transposed), None
    outs = _SparseConv.apply(feats, weight, entry, transposed, bool(bn_stats), bool(with_skip))
    outs = outs if isinstance(outs, tuple) else (outs,)
    out = outs[0]
    sums = outs[1] if bn_stats else None
    res = (out, (sums if sums is not None and sums.numel() else None))
    return res + (outs[-1],) if with_skip else res


def prebuild_coords(x, steps):
    """Output coordinates of the strided convolutions a network is KNOWN to apply to x's coordinate set, in order -- steps =
    [(stride, kernel_size), ...] -- computed now and left in x.cmaps, where conv3d finds them (the same dict entry the first
    strided convolution of each level would have made: TS:torchsparse/nn/functional/conv.py:156-164). Every spdownsample ends in
    a host read of its output size; issued lazily those reads sit between the encoder stages, each one draining the launch queue
    the host had built up. Up front they cost the same device work and leave the rest of forward + backward without a read."""
    coords, ts = x.coords, make_ntuple(x.stride, ndim=3)
    for stride, kernel_size in steps:
        stride, kernel_size = make_ntuple(stride, ndim=3), make_ntuple(kernel_size, ndim=3)
        out_stride = tuple(ts[k] * stride[k] for k in range(3))
        if out_stride in x.cmaps:
            coords = x.cmaps[out_stride]
        elif all(s == 1 for s in stride):
            pass
        else:
            coords = spdownsample(coords, stride, kernel_size, ts)
            x.cmaps[out_stride] = coords
        ts = out_stride
    return x


def conv3d(input, weight, kernel_size, bias=None, stride=1, dilation=1, transposed=False, bn_stats=False, with_skip=False,
           act_slope=None):
    """bn_stats (not in the reference's signature; used by the fused blocks): ask the convolution for the BatchNorm
    statistics of its output; they are attached to the returned tensor as `.bn_sums` when the kernel produced them.
    with_skip (likewise): return (output, skip) where skip is the INPUT tensor again, routed through the convolution's autograd
    node: a caller that also feeds the input to a residual / skip path uses `skip` there, and the two gradients of the input
    are summed inside the dgrad kernel's write-back (see _SparseConv)."""
    skip_feats = None
    kernel_size = make_ntuple(kernel_size, ndim=3)
    stride = make_ntuple(stride, ndim=3)
    dilation = make_ntuple(dilation, ndim=3)
    ones = (1, 1, 1)
    bn_sums = None

    if kernel_size == ones and stride == ones and dilation == ones:
        output_stride = input.stride
        output_coords = input.coords
        if input.feats.is_cuda and weight.dim() == 2:
            output_feats = _PointwiseConv.apply(input.feats, weight, input.kmaps)
        else:
            output_feats = input.feats.matmul(weight)
    elif not transposed:
        output_stride = tuple(input.stride[k] * stride[k] for k in range(3))
        if output_stride in input.cmaps:
            output_coords = input.cmaps[output_stride]
        elif all(stride[k] == 1 for k in range(3)):
            output_coords = input.coords
        else:
            output_coords = spdownsample(input.coords, stride, kernel_size, input.stride)
        key = (input.stride, kernel_size, stride, dilation)
        if key not in input.kmaps:
            input.kmaps[key] = build_kernel_map(input.coords, output_coords, kernel_size,
                                                input.stride, dilation)
        if with_skip:
            output_feats, bn_sums, skip_feats = _sparse_conv(input.feats, weight, input.kmaps[key], False, bn_stats and bias is None, True, act_slope)
        else:
            output_feats, bn_sums = _sparse_conv(input.feats, weight, input.kmaps[key], False, bn_stats and bias is None, False, act_slope)
    else:
        output_stride = tuple(input.stride[k] // stride[k] for k in range(3))
        output_coords = input.cmaps[output_stride]
        key = (output_stride, kernel_size, stride, dilation)
        if with_skip:
            output_feats, bn_sums, skip_feats = _sparse_conv(input.feats, weight, input.kmaps[key], True, bn_stats and bias is None, True, act_slope)
        else:
            output_feats, bn_sums = _sparse_conv(input.feats, weight, input.kmaps[key], True, bn_stats and bias is None, False, act_slope)

    if bias is not None:
        output_feats += bias

    output = SparseTensor(coords=output_coords, feats=output_feats, stride=output_stride)
    output.cmaps = input.cmaps
    output.cmaps.setdefault(output_stride, output_coords)
    output.kmaps = input.kmaps
    if bn_sums is not None:
        # bound to the feature tensor they describe: FusedBatchNorm ignores them if feats was replaced or modified
        try:
            output.bn_sums = (bn_sums, output_feats, output_feats._version)
        except RuntimeError:  # inference tensors track no version counter: the BatchNorm runs its own statistics pass
            pass
    if with_skip:
        return output, input._like(skip_feats if skip_feats is not None else input.feats)
    return output


# ---------------------------------------------------------------------------------------------
def relu(input, inplace=True):
    return input._like(torch.nn.functional.relu(input.feats, inplace=inplace))


def leaky_relu(input, negative_slope=0.1, inplace=True):
    return input._like(torch.nn.functional.leaky_relu(input.feats, negative_slope, inplace=inplace))
