"""`range_utils.nn.functional.{map_count, denselize}` of the reference's range_lib
(RL = R:pcseg/model/segmentor/fusion/rpvnet/range_lib/; RL:range_utils/nn/functional/map_count.py:7-28,
denselize.py:7-34; called from R:pcseg/model/segmentor/fusion/rpvnet/rpvnet.py:73-91), on the HIP backend -- and the opposite
direction, `range_to_point` (rpvnet.py:31-51: grid_sample per frame), as one launch each way."""
import sys
import types

import torch
from torch.amp import custom_bwd, custom_fwd
from torch.autograd import Function

from . import native


def map_count(pxpy, max_bs, h, w):
    """pxpy (N,3) int32 rows (batch, px, py) -> (B,H,W) int32 counts."""
    return native.backend().map_count(pxpy.contiguous().int(), int(max_bs), int(h), int(w))


class _Denselize(Function):
    @staticmethod
    def forward(ctx, feat, count_map, pxpy):
        count_map = count_map.int().contiguous()
        pxpy = pxpy.int().contiguous()
        out = native.backend().denselize_fwd(feat.float().contiguous(), count_map, pxpy)
        ctx.for_backwards = (count_map, pxpy)
        return out

    @staticmethod
    def backward(ctx, top_grad):
        count_map, pxpy = ctx.for_backwards
        return native.backend().denselize_bwd(top_grad.float().contiguous(), count_map, pxpy), None, None


def denselize(feat, count_map, pxpy):
    """scatter-mean of point features (N,C) into a (B,C,H,W) range image."""
    return _Denselize.apply(feat, count_map, pxpy)


class _RangeToPoint(Function):
    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)   # grid_sampler is an fp32 op under autocast
    def forward(ctx, feature_map, pxpy):
        feature_map = feature_map.contiguous()
        pxpy = pxpy.contiguous()
        ctx.for_backwards = (pxpy, tuple(feature_map.shape))
        return native.backend().range_sample_fwd(feature_map, pxpy)

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, grad_out):
        pxpy, (b, c, h, w) = ctx.for_backwards
        if _PENDING:
            verify_pending()
        return native.backend().range_sample_bwd(grad_out.float().contiguous(), pxpy, b, h, w), None


# The kernels return the rows in INPUT order; the reference's per-frame loop (rpvnet.py:36-50) returns them grouped by ascending
# frame, which is the same thing exactly when the frame column is non-decreasing integers in [0, b) -- what a collated batch is.
# Round 5 read that flag back with `.item()`: one device synchronisation per new pxpy tensor, i.e. per training step. Now the flag
# is computed on the device, copied to pinned host memory behind the kernels, and VERIFIED LATER (at the next call of this module,
# in backward, or by `verify_pending()`): a violation raises instead of returning rows in an order the reference would not.
_PENDING = []       # (event, pinned flag, shape) of checks whose copy has not been looked at yet
_FLAG_POOL = []


def verify_pending(block=False):
    """Look at the frame-order checks whose result has arrived (block=True: wait for all of them). Raises RuntimeError when a
    `range_to_point` input had its frames out of order: its result was in input order, the reference's is grouped by frame."""
    keep, bad = [], None
    for ev, flag, shape in _PENDING:
        if block:
            ev.synchronize()
        if ev.query():
            if not bool(flag.item()):
                bad = shape
            _FLAG_POOL.append(flag)
        else:
            keep.append((ev, flag, shape))
    _PENDING[:] = keep
    if bad is not None:
        raise RuntimeError("openpcseg_amd.rangelib.range_to_point: a pxpy tensor %s had its frame column out of order (not non-decreasing "
                           "integers in [0, B)); the fused kernels returned its rows in input order, the reference regroups them by "
                           "frame. Sort the points by frame, or call the reference's function for such inputs." % (bad,))


def _frames_in_order(pxpy, b):
    """True when the frame column is non-decreasing integers in [0, b). Device tensors: assumed, and verified asynchronously
    (see above); host tensors: checked at once."""
    def check():
        f = pxpy[:, 0]
        ok = ((f >= 0) & (f < b) & (f == torch.floor(f))).all() & (f[1:] >= f[:-1]).all()
        if not pxpy.is_cuda:
            return bool(ok.item())
        flag = _FLAG_POOL.pop() if _FLAG_POOL else torch.empty((), dtype=torch.bool).pin_memory()
        flag.copy_(ok, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        _PENDING.append((ev, flag, tuple(pxpy.shape)))
        return True
    if _PENDING:
        verify_pending()
    return native._cached(pxpy, "_pcs_frames_sorted", native._cache_key(pxpy) + (b,), check)


def range_to_point(feature_map, pxpy, grid_sample_mode="bilinear", fallback=None):
    """(N, C) features of the points: bilinear samples of the (B, C, H, W) range feature map at pxpy (N, 3) = (frame, x, y) --
    R:pcseg/model/segmentor/fusion/rpvnet/rpvnet.py:31-51 without the loop over frames. Inputs the kernels do not serve (another
    sampling mode, channel counts that are not a multiple of 4, host tensors) go to `fallback` (the reference's own function) when
    one is given. Rows whose frames are not grouped in ascending order -- the reference would REORDER those -- are detected on the
    device without a host read and raise at the next `verify_pending()` point (see `_frames_in_order`)."""
    # the kernels take float32 (any float16 / bfloat16 input only under CUDA autocast, whose custom_fwd casts to float32)
    f32 = (feature_map.dtype == torch.float32 and pxpy.dtype == torch.float32) or (
        torch.is_autocast_enabled("cuda") and feature_map.dtype in (torch.float32, torch.float16, torch.bfloat16) and
        pxpy.dtype in (torch.float32, torch.float16, torch.bfloat16))
    ok = (grid_sample_mode == "bilinear" and feature_map.is_cuda and feature_map.dim() == 4 and feature_map.shape[1] % 4 == 0 and
          pxpy.dim() == 2 and pxpy.shape[1] == 3 and pxpy.is_floating_point() and pxpy.shape[0] > 0 and f32 and
          _frames_in_order(pxpy, feature_map.shape[0]))
    if not ok:
        if fallback is None:
            raise RuntimeError("openpcseg_amd.rangelib.range_to_point: unsupported input (mode %r, feature map %s, pxpy %s)" %
                               (grid_sample_mode, tuple(feature_map.shape), tuple(pxpy.shape)))
        return fallback(feature_map, pxpy, grid_sample_mode)
    return _RangeToPoint.apply(feature_map, pxpy)


def point_to_range(pf, pxpy, b, h, w):
    """(B, C, H, W) range feature map from point features: R:pcseg/model/segmentor/fusion/rpvnet/rpvnet.py:73-91 -- pixel of a point =
    trunc((p + 1) / 2 * (size - 1)) per axis, then map_count + denselize -- without its `torch.Tensor([w-1, h-1]).cuda()` (a pageable
    host-to-device copy: a full device synchronisation, four times per RPVNet step). Same float32 arithmetic per element (the reference
    broadcasts a (2,) float32 tensor, here each column meets its python scalar), so the integer pixels are identical. The integer
    coordinates are cached on the pxpy tensor per resolution: forward, backward and the next call on the same tensor share one pixel CSR."""
    def make():
        p32 = pxpy.float() if pxpy.dtype in (torch.float16, torch.bfloat16) else pxpy   # the reference's float32 factor promotes halfs
        px = (p32[:, 1] + 1) / 2 * float(w - 1)
        py = (p32[:, 2] + 1) / 2 * float(h - 1)
        return torch.stack([p32[:, 0], px, py], dim=1).int().contiguous()
    int_pxpy = native._cached(pxpy, "_pcs_int_pxpy_%dx%d" % (h, w), native._cache_key(pxpy), make)
    return denselize(pf, map_count(int_pxpy, b, h, w), int_pxpy)


def install_as_range_utils():
    fn = types.ModuleType("range_utils.nn.functional")
    fn.map_count, fn.denselize = map_count, denselize
    nn = types.ModuleType("range_utils.nn")
    nn.functional = fn
    top = types.ModuleType("range_utils")
    top.nn = nn
    top.__path__, nn.__path__ = [], []
    top.__openpcseg_amd__ = True
    for m in (top, nn, fn):
        sys.modules[m.__name__] = m
    return top
