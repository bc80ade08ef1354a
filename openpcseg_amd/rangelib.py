"""`range_utils.nn.functional.{map_count, denselize}` of the reference's range_lib
(RL = R:pcseg/model/segmentor/fusion/rpvnet/range_lib/; RL:range_utils/nn/functional/map_count.py:7-28,
denselize.py:7-34; called from R:pcseg/model/segmentor/fusion/rpvnet/rpvnet.py:73-91), on the HIP backend."""
import sys
import types

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
