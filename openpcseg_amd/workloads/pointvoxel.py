"""Point <-> voxel glue used by the MinkUNet workload (the op sequences of SURVEY.md 3.2).

Behaviourally equivalent to R:pcseg/model/segmentor/voxel/minkunet/utils.py:11-105
(initial_voxelize / voxel_to_point), written against this package's functional API. The
reference's own copy runs unmodified on top of `install_as_torchsparse()` as well (tested
in-container); this copy exists because the reference tree does not travel to the GPU box.
"""
import os

import torch

from .. import functional as F
from .. import native
from ..sparse import PointTensor, SparseTensor, get_kernel_offsets


VOXEL_ORDER = os.environ.get("PCS_VOXEL_ORDER", "hash")  # "hash" (the reference's row order) | "spatial" (experiment)


def initial_voxelize(z, init_res, after_res):
    """Points -> stride-1 voxels ordered by ascending 60-bit hash (utils.py:11-36). That order scatters spatial
    neighbours over the whole tensor (every row a stride-1 convolution gathers is its own fetch: 3.7x the algorithmic
    bytes at the fabric counters). PCS_VOXEL_ORDER=spatial orders the voxels by (batch, x, y, z) instead -- same voxel
    set, same per-point results. Measured (12-frame batch, fp32 and bf16 steps): no difference, 88.7 vs 88.2 and 159.8
    vs 160.1 frames/s -- the gathers of a 220-450 MB level are served by the 256 MB Infinity Cache either way -- so the
    reference's order stays the default."""
    fc = torch.cat([(z.C[:, :3] * init_res) / after_res, z.C[:, -1:].clone()], dim=1)
    cell = torch.floor(fc)
    icell = cell.int()
    pc_hash = F.sphash(icell)
    be = native.backend()
    if VOXEL_ORDER == "spatial" and icell.is_cuda and hasattr(be, "downsample"):
        voxel_hash = F.sphash(be.downsample(icell.contiguous(), [1, 1, 1]))  # unique cells in (b, x, y, z) order
        idx_query = F.sphashquery(pc_hash, voxel_hash)
        counts = F.spcount(idx_query.int(), len(voxel_hash))
    elif icell.is_cuda and hasattr(be, "unique_inverse_csr") and os.environ.get("PCS_FUSED_VOXELIZE", "1") != "0":
        # torch.unique + sphashquery + spcount of the reference (utils.py:17-19) from ONE stable sort of the point
        # hashes: unique hashes (ascending), point -> voxel map, counts; the sort is also the CSR of the two
        # segmented spvoxelize passes below (no table build + probe, no second sort)
        voxel_hash, idx_query, counts = be.unique_inverse_csr(pc_hash)
    else:
        voxel_hash = torch.unique(pc_hash)
        idx_query = F.sphashquery(pc_hash, voxel_hash)
        counts = F.spcount(idx_query.int(), len(voxel_hash))
    coords = torch.round(F.spvoxelize(cell, idx_query, counts)).int()
    feats = F.spvoxelize(z.F, idx_query, counts)
    x = SparseTensor(feats, coords, 1)
    x.cmaps.setdefault(x.stride, x.coords)
    z.additional_features["idx_query"][1] = idx_query
    z.additional_features["counts"][1] = counts
    z.C = fc
    return x


def point_to_voxel(x, z):
    """Mean of the point features of z in the voxels of x (utils.py:41-64); the point -> voxel map is cached on z per stride."""
    af = z.additional_features
    if af is None or af.get("idx_query") is None or af["idx_query"].get(x.s) is None:
        cell = torch.cat([torch.floor(z.C[:, :3] / x.s[0]).int() * x.s[0], z.C[:, -1].int().view(-1, 1)], 1)
        pc_hash = F.sphash(cell)
        be = native.backend()
        if hasattr(be, "level_table") and x.C.is_cuda:
            idx_query = be.table_query(be.level_table(x.C), pc_hash) - 1   # the level's cached table (shared with its kernel maps)
        else:
            idx_query = F.sphashquery(pc_hash, F.sphash(x.C))
        counts = F.spcount(idx_query.int(), x.C.shape[0])
        af["idx_query"][x.s] = idx_query
        af["counts"][x.s] = counts
    else:
        idx_query, counts = af["idx_query"][x.s], af["counts"][x.s]
    out = SparseTensor(F.spvoxelize(z.F, idx_query, counts), x.C, x.s)
    out.cmaps, out.kmaps = x.cmaps, x.kmaps
    return out


def point_maps(x, z, nearest=False):
    """(idx_query (N, 8), weights (N, 8)) of the points of z in the voxels of x; cached on z per stride (utils.py:69-105)."""
    s = x.s
    if z.idx_query.get(s) is None or z.weights.get(s) is None:
        be = native.backend()
        if hasattr(be, "corner_map") and z.C.is_cuda and not nearest and os.environ.get("PCS_CORNER_MAP", "1") != "0":
            # the whole map in one kernel: corner hashes, table lookup (the level's cached table), trilinear weights
            idx_query, weights = be.corner_map(z.C, x.C, s[0])
        else:
            corners = get_kernel_offsets(2, s, 1, device=z.F.device)
            base = torch.cat([torch.floor(z.C[:, :3] / s[0]).int() * s[0], z.C[:, -1:].int()], dim=1)
            idx_query = F.sphashquery(F.sphash(base, corners), F.sphash(x.C.to(z.F.device)))
            weights = F.calc_ti_weights(z.C, idx_query, scale=s[0]).t().contiguous()
            idx_query = idx_query.t().contiguous()
        if nearest:
            weights[:, 1:] = 0.0
            idx_query[:, 1:] = -1
        z.idx_query[s] = idx_query
        z.weights[s] = weights
    return z.idx_query[s], z.weights[s]


def voxel_to_point(x, z, nearest=False):
    """Trilinear devoxelisation of x at the points of z; maps cached per stride (utils.py:69-105)."""
    s = x.s
    point_maps(x, z, nearest)
    out = PointTensor(F.spdevoxelize(x.F, z.idx_query[s], z.weights[s]), z.C,
                      idx_query=z.idx_query, weights=z.weights)
    out.additional_features = z.additional_features
    return out
