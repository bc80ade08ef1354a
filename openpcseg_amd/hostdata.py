"""Voxel dedup and batch collation: NumPy in the dataloader workers like the reference (SURVEY.md section 8
a15), and -- for scans already resident in HBM (a device tensor in) -- on the MI355X through the pcs_quantize_*
kernels (section 8 f1). Same signatures and ordering contract as
  sparse_quantize / ravel_hash   TS:torchsparse/utils/quantize.py:9-46
  sparse_collate(_fn)            TS:torchsparse/utils/collate.py:11-59
"""
from itertools import repeat

import numpy as np
import torch

from .sparse import SparseTensor

__all__ = ["ravel_hash", "sparse_quantize", "sparse_quantize_frames", "sparse_collate", "sparse_collate_fn"]


def ravel_hash(x):
    """Row-major linear index of integer coords inside their bounding box (uint64)."""
    assert x.ndim == 2, x.shape
    x = (x - x.min(axis=0)).astype(np.uint64, copy=False)
    extent = x.max(axis=0).astype(np.uint64) + 1
    h = np.zeros(x.shape[0], dtype=np.uint64)
    for d in range(x.shape[1] - 1):
        h += x[:, d]
        h *= extent[d + 1]
    h += x[:, -1]
    return h


def sparse_quantize(coords, voxel_size=1, *, return_index=False, return_inverse=False):
    """floor(coords / voxel_size) -> unique voxels, one representative (first occurrence) per
    voxel, output ordered by ascending ravel hash."""
    if isinstance(voxel_size, (float, int)):
        voxel_size = tuple(repeat(voxel_size, 3))
    assert isinstance(voxel_size, tuple) and len(voxel_size) == 3
    if isinstance(coords, torch.Tensor):  # device tensor in -> device tensors out (no CPU path for tensors)
        from . import native
        vox, index, inverse = native.backend().quantize(coords, voxel_size, return_index, return_inverse)
        outputs = [vox] + ([index] if return_index else []) + ([inverse] if return_inverse else [])
        return outputs[0] if len(outputs) == 1 else outputs
    coords = np.floor(coords / np.array(voxel_size)).astype(np.int32)
    _, indices, inverse = np.unique(ravel_hash(coords), return_index=True, return_inverse=True)
    outputs = [coords[indices]]
    if return_index:
        outputs.append(indices)
    if return_inverse:
        outputs.append(inverse)
    return outputs[0] if len(outputs) == 1 else outputs


def sparse_quantize_frames(coords, frames, num_frames):
    """`sparse_quantize` (voxel size 1, integer coordinates) of EVERY frame of a batch plus `sparse_collate`'s batch column in one
    device pass: coords (N, 3) int32 on the device, frames (N,) frame of each row, ascending.
    Returns (voxels (M, 4) int32 [x, y, z, frame], index (M,) int64 -- the representative row of each voxel --, inverse (N,) int64).
    Per frame exactly what TS:torchsparse/utils/quantize.py:24-46 returns for that frame's rows -- voxels ordered by ascending
    ravel hash inside the frame's bounding box, first occurrence as representative -- and the frames concatenated in order like
    TS:torchsparse/utils/collate.py:11-32. One stable radix sort and one host read (the voxel count sizes the outputs) for the
    whole batch, where the per-frame form costs a sort, a scan and a host read per frame.
    Range: the key is an int64, so num_frames x (the batch's bounding-box volume) must stay below 2^63 -- the reference's per-frame
    uint64 ravel hash has the same kind of limit per frame (a 12-frame SemanticKITTI batch: 12 x 1035 x 1094 x 56 = 7.6e8)."""
    from . import native
    be = native.backend()
    if not (isinstance(coords, torch.Tensor) and coords.is_cuda and coords.dim() == 2 and coords.shape[1] == 3):
        raise RuntimeError("sparse_quantize_frames: coords must be an (N, 3) tensor on the HIP device")
    coords = coords.int().contiguous()
    f = frames.long()
    # ravel_hash (quantize.py:15-21) = ((x - x0) ey + (y - y0)) ez + (z - z0) inside the frame's bounding box: ascending ravel hash IS
    # ascending lexicographic (x, y, z) -- so one key with the frame on top and the BATCH's bounding box below orders every frame
    # like its own ravel hash does, without any per-frame quantity (12-way atomic min / max over 1.4 M rows: 200 ms)
    if hasattr(be, "quantize_frame_keys"):
        key = be.quantize_frame_keys(coords, f)     # one launch (round 5: a dozen torch elementwise launches)
    else:
        lo = coords.amin(0).long()
        ext = coords.amax(0).long() - lo + 1
        c = coords.long() - lo
        key = ((f * ext[0] + c[:, 0]) * ext[1] + c[:, 1]) * ext[2] + c[:, 2]
    return be.quantize_sorted_keys(key, coords, f)


def sparse_collate(inputs):
    """Concatenate SparseTensors along N, appending the batch index as the 4th coord column."""
    stride = inputs[0].stride
    coords, feats = [], []
    for b, x in enumerate(inputs):
        if isinstance(x.coords, np.ndarray):
            x.coords = torch.tensor(x.coords)
        if isinstance(x.feats, np.ndarray):
            x.feats = torch.tensor(x.feats)
        assert isinstance(x.coords, torch.Tensor), type(x.coords)
        assert isinstance(x.feats, torch.Tensor), type(x.feats)
        assert x.stride == stride, (x.stride, stride)
        col = torch.full((x.coords.shape[0], 1), b, device=x.coords.device, dtype=torch.int)
        coords.append(torch.cat((x.coords, col), dim=1))
        feats.append(x.feats)
    return SparseTensor(coords=torch.cat(coords, dim=0), feats=torch.cat(feats, dim=0), stride=stride)


def sparse_collate_fn(inputs):
    if not isinstance(inputs[0], dict):
        return inputs
    out = {}
    for name, first in inputs[0].items():
        column = [sample[name] for sample in inputs]
        if isinstance(first, dict):
            out[name] = sparse_collate_fn(column)
        elif isinstance(first, np.ndarray):
            out[name] = torch.stack([torch.tensor(v) for v in column], dim=0)
        elif isinstance(first, torch.Tensor):
            out[name] = torch.stack(column, dim=0)
        elif isinstance(first, SparseTensor):
            out[name] = sparse_collate(column)
        else:
            out[name] = column
    return out
