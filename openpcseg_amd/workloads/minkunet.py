"""MinkUNet (ResBlock variant) workload on the HIP operator API.

Architecture and state_dict layout of R:pcseg/model/segmentor/voxel/minkunet/minkunet.py:194-434
(so reference checkpoints load): stem 2x[k3 conv, BN, ReLU]; 4 encoder stages
[k2 s2 conv-BN-ReLU + n residual blocks]; 4 decoder stages [k2 s2 transposed conv-BN-ReLU,
concat skip, n residual blocks]; point branch = trilinear devoxelise at strides 1/16/4/1;
Linear classifier over the concatenated point features. mk34 = NUM_LAYER [2,3,4,6,2,2,2,2],
mk18 = [2,2,2,2,2,2,2,2]; PLANES [32,32,64,128,256,256,128,96,96] x cr.
"""
import os

import torch
from torch import nn

from .. import functional as F
from .. import modules as spnn
from ..fused import FusedBatchNorm, FusedLinear
from ..sparse import PointTensor, cat, fapply
from .losses import SegLoss
from .pointvoxel import initial_voxelize, point_maps, voxel_to_point

MK34_LAYERS = [2, 3, 4, 6, 2, 2, 2, 2]
MK18_LAYERS = [2, 2, 2, 2, 2, 2, 2, 2]
PLANES = [32, 32, 64, 128, 256, 256, 128, 96, 96]


class _BN(nn.BatchNorm1d):
    def forward(self, x):
        return fapply(x, super().forward)


class _SyncBN(nn.SyncBatchNorm):
    def forward(self, x):
        return fapply(x, super().forward)


FUSED = True  # BatchNorm + residual + ReLU as fused HIP passes (openpcseg_amd.fused); False = torch modules


def _norm(c, dist):
    if FUSED:
        return FusedBatchNorm(c, sync=dist)
    return _SyncBN(c) if dist else _BN(c)


def _link(conv, bn):
    """A convolution that feeds a FusedBatchNorm hands over the statistics of its output (conv write-back)."""
    if isinstance(bn, FusedBatchNorm) and isinstance(conv, spnn.Conv3d):
        conv.emit_bn_stats = os.environ.get("PCS_CONV_BN_STATS", "1") != "0"


CAT_FUSED = os.environ.get("PCS_CAT_FUSED", "1") != "0"  # decoder concat written by the BN apply pass
PREBUILD = os.environ.get("PCS_PREBUILD_LEVELS", "1") != "0"   # all levels' coordinates at the start of forward [r6]
SKIP_FUSED = os.environ.get("PCS_SKIP_FUSED", "1") != "0"  # residual-block skip gradient added in the dgrad write-back [r6]


def _bn_act(bn, x, residual=None, relu=True, act=None, cat_with=None):
    """BN (+residual) (+ReLU) (+concat with the skip tensor): one fused pass, or the reference's separate torch modules."""
    if isinstance(bn, FusedBatchNorm):
        if cat_with is not None and not CAT_FUSED:
            return cat([bn(x, residual=residual, relu=relu), cat_with])
        return bn(x, residual=residual, relu=relu, cat_with=cat_with)
    y = bn(x)
    if residual is not None:
        y = y + residual
    y = act(y) if relu else y
    return y if cat_with is None else cat([y, cat_with])


class ConvBlock(nn.Module):
    """conv -> BN -> ReLU (`net.0`, `net.1`); transposed=True gives the decoder up-conv."""

    def __init__(self, cin, cout, ks, stride, dist, transposed=False):
        super().__init__()
        self.net = nn.Sequential(spnn.Conv3d(cin, cout, kernel_size=ks, stride=stride, transposed=transposed),
                                 _norm(cout, dist), spnn.ReLU(True))
        _link(self.net[0], self.net[1])

    def forward(self, x, cat_with=None):
        return _bn_act(self.net[1], self.net[0](x), act=self.net[2], cat_with=cat_with)


class ResBlock(nn.Module):
    def __init__(self, cin, cout, dist):
        super().__init__()
        self.net = nn.Sequential(spnn.Conv3d(cin, cout, kernel_size=3), _norm(cout, dist), spnn.ReLU(True),
                                 spnn.Conv3d(cout, cout, kernel_size=3), _norm(cout, dist))
        if cin == cout:
            self.downsample = nn.Identity()
        else:
            self.downsample = nn.Sequential(spnn.Conv3d(cin, cout, kernel_size=1), _norm(cout, dist))
        self.relu = spnn.ReLU(True)
        _link(self.net[0], self.net[1])
        _link(self.net[3], self.net[4])

    def forward(self, x):
        if SKIP_FUSED and torch.is_grad_enabled() and x.feats.requires_grad:
            # x feeds the first convolution AND the skip path: both through the convolution's autograd node, whose dgrad kernel
            # adds the skip gradient in its write-back (no elementwise sum of two gradient tensors per block)
            h0, xs = self.net[0](x, with_skip=True)
        else:
            h0, xs = self.net[0](x), x
        h = _bn_act(self.net[1], h0, act=self.net[2])
        if isinstance(self.downsample, nn.Identity):
            r = xs
        else:
            r = _bn_act(self.downsample[1], self.downsample[0](xs), relu=False)
        return _bn_act(self.net[4], self.net[3](h), residual=r, act=self.relu)


def _res_stack(cin, cout, n, dist):
    return [ResBlock(cin if i == 0 else cout, cout, dist) for i in range(n)]


class MinkUNet(nn.Module):
    def __init__(self, num_class=20, in_dim=4, num_layer=MK34_LAYERS, planes=PLANES, cr=1.0,
                 pres=0.05, vres=0.05, dist=False, ignore_label=0, label_smoothing=0.1, dropout=0.0):
        super().__init__()
        cs = [int(cr * c) for c in planes]
        self.in_dim, self.pres, self.vres = in_dim, pres, vres
        self.stem = nn.Sequential(spnn.Conv3d(in_dim, cs[0], kernel_size=3), _norm(cs[0], dist), spnn.ReLU(True),
                                  spnn.Conv3d(cs[0], cs[0], kernel_size=3), _norm(cs[0], dist), spnn.ReLU(True))
        _link(self.stem[0], self.stem[1])
        _link(self.stem[3], self.stem[4])
        enc_in = [cs[0], cs[1], cs[2], cs[3]]
        for i in range(4):
            setattr(self, "stage%d" % (i + 1), nn.Sequential(
                ConvBlock(enc_in[i], enc_in[i], 2, 2, dist), *_res_stack(enc_in[i], cs[i + 1], num_layer[i], dist)))
        skip = [cs[3], cs[2], cs[1], cs[0]]
        cin = cs[4]
        for i in range(4):
            cout = cs[5 + i]
            setattr(self, "up%d" % (i + 1), nn.ModuleList([
                ConvBlock(cin, cout, 2, 2, dist, transposed=True),
                nn.Sequential(*_res_stack(cout + skip[i], cout, num_layer[4 + i], dist))]))
            cin = cout
        self.classifier = nn.Sequential((FusedLinear if FUSED else nn.Linear)(cs[4] + cs[6] + cs[8], num_class))
        self.dropout = nn.Dropout(dropout, True)
        self.criterion = SegLoss(ignore_index=ignore_label, label_smoothing=label_smoothing)
        # num_batches_tracked of the 63 BatchNorm layers: one _foreach_add_ per training step instead of 63 scalar kernels
        self._bn_layers = [m for m in self.modules() if isinstance(m, FusedBatchNorm)]
        for m in self._bn_layers:
            m.counted_by_parent = True

    def _stem(self, x):
        h = _bn_act(self.stem[1], self.stem[0](x), act=self.stem[2])
        return _bn_act(self.stem[4], self.stem[3](h), act=self.stem[5])

    def _dropout(self, feats, out_of_place):
        if out_of_place:
            return torch.nn.functional.dropout(feats, self.dropout.p, self.training, False)
        return self.dropout(feats)

    def point_logits(self, x):
        """x: SparseTensor (feats (N,>=in_dim), coords (N,4) int) -> per-point logits (N, num_class)."""
        if self.training and self._bn_layers:
            torch._foreach_add_([m.num_batches_tracked for m in self._bn_layers], 1)
        x.F = x.F[:, :self.in_dim]
        z = PointTensor(x.F, x.C.float())
        xv = initial_voxelize(z, self.pres, self.vres)
        if PREBUILD:   # the four stride-2 levels now: their size read-backs leave the encoder (functional.prebuild_coords)
            F.prebuild_coords(xv, [(2, 2)] * 4)
        x0 = self._stem(xv)
        z0 = voxel_to_point(x0, z)
        lin = self.classifier[0]
        # classifier(cat(devoxelize(x4), devoxelize(y2), devoxelize(y4))) with the linear map applied on the voxels first
        # (FusedLinear.devoxelized_part: interpolation and the classifier commute); PCS_CLASSIFIER_COMMUTE=0 = the literal
        # order. The voxel features are taken where the reference devoxelises them: BEFORE the dropout that follows (which
        # then runs out of place: the product keeps its input for the weight gradient).
        commute = isinstance(lin, FusedLinear) and os.environ.get("PCS_CLASSIFIER_COMMUTE", "1") != "0"
        x1 = self.stage1(x0)
        x2 = self.stage2(x1)
        x3 = self.stage3(x2)
        x4 = self.stage4(x3)
        if commute:
            t1 = lin.devoxelized_part(0, x4.F, *point_maps(x4, z0), cache=x4.kmaps)
        else:
            z1 = voxel_to_point(x4, z0)
        x4.F = self._dropout(x4.F, commute)
        y1 = self.up1[1](self.up1[0](x4, cat_with=x3))  # torchsparse.cat([up(x4), x3]) fused into the BN apply pass
        y2 = self.up2[1](self.up2[0](y1, cat_with=x2))
        if commute:
            t2 = lin.devoxelized_part(x4.F.shape[1], y2.F, *point_maps(y2, z0), cache=y2.kmaps)
        else:
            z2 = voxel_to_point(y2, z1)
        y2.F = self._dropout(y2.F, commute)
        y3 = self.up3[1](self.up3[0](y2, cat_with=x1))
        y4 = self.up4[1](self.up4[0](y3, cat_with=x0))
        if commute:
            t3 = lin.devoxelized_part(x4.F.shape[1] + y2.F.shape[1], y4.F, *point_maps(y4, z0), cache=y4.kmaps)
            return lin.sum_devoxelized([t1, t2, t3])
        z3 = voxel_to_point(y4, z2)
        if isinstance(lin, FusedLinear) and os.environ.get("PCS_CLASSIFIER_PARTS", "1") != "0":
            return lin.forward_parts([z1.F, z2.F, z3.F])  # Linear over [z1 | z2 | z3] without the (N, 480) concat
        return self.classifier(torch.cat([z1.F, z2.F, z3.F], dim=1))

    def forward(self, batch):
        logits = self.point_logits(batch["lidar"])
        out = {"logits": logits}
        if self.training and "targets" in batch:
            out["loss"] = self.criterion(logits, batch["targets"].F.long())
        return out
