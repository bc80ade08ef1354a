"""Voxel dedup and batch collation: NumPy in the dataloader workers like the reference (SURVEY.md section 8
a15), and -- for scans already resident in HBM (a device tensor in) -- on the MI355X through the pcs_quantize_*
kernels (section 8 f1). Same signatures and ordering contract as
  sparse_quantize / ravel_hash   TS:torchsparse/utils/quantize.py:9-46
  sparse_collate(_fn)            TS:torchsparse/utils/collate.py:11-59
"""
import numpy as np
import torch

from .sparse import SparseTensor

__all__ = ["ravel_hash", "sparse_quantize", "sparse_quantize_frames", "sparse_collate", "sparse_collate_fn"]


def ravel_hash(x):
    """Row-major linear index of integer coordinate rows inside their own bounding box, uint64: the sort key whose ascending
    order is the output order of `sparse_quantize` (TS:torchsparse/utils/quantize.py:12-21 defines that order)."""
    x = np.asarray(x)
    if x.ndim != 2:
        raise AssertionError(x.shape)
    if x.shape[0] == 0:
        return np.zeros(0, dtype=np.uint64)
    rel = x.astype(np.int64) - x.min(axis=0).astype(np.int64)
    extent = rel.max(axis=0) + 1
    if float(np.prod(extent.astype(np.float64))) < 2.0 ** 62:   # the usual case: one vectorised mixed-radix evaluation
        return np.ravel_multi_index(tuple(rel.T), tuple(int(e) for e in extent)).astype(np.uint64)
    key = rel[:, 0].astype(np.uint64)                            # huge boxes: the same index modulo 2^64
    for d in range(1, x.shape[1]):
        key = key * np.uint64(extent[d]) + rel[:, d].astype(np.uint64)
    return key


def _first_of_each_run(key):
    """One stable sort of the keys -> (rows of the FIRST occurrence of every distinct key, in ascending key order;
    for every row the rank of its key). The same three steps the device path runs (sort, run flags, scan:
    csrc/quantize.hip), instead of np.unique's bookkeeping."""
    n = key.shape[0]
    order = np.argsort(key, kind="stable")
    sk = key[order]
    starts = np.ones(n, dtype=bool)
    starts[1:] = sk[1:] != sk[:-1]
    rank = np.cumsum(starts) - 1
    inverse = np.empty(n, dtype=np.int64)
    inverse[order] = rank
    return order[starts], inverse


def sparse_quantize(coords, voxel_size=1, *, return_index=False, return_inverse=False):
    """floor(coords / voxel_size) -> the distinct voxels, each represented by its first row, ordered by ascending ravel hash
    (same signature, outputs and order as TS:torchsparse/utils/quantize.py:24-46)."""
    if isinstance(voxel_size, (float, int)):
        voxel_size = (voxel_size,) * 3
    if not (isinstance(voxel_size, tuple) and len(voxel_size) == 3):
        raise AssertionError(voxel_size)
    if isinstance(coords, torch.Tensor):  # device tensor in -> device tensors out (no CPU path for tensors)
        from . import native
        vox, index, inverse = native.backend().quantize(coords, voxel_size, return_index, return_inverse)
    else:
        cells = np.floor(coords / np.array(voxel_size)).astype(np.int32)
        index, inverse = _first_of_each_run(ravel_hash(cells))
        vox = cells[index]
    extras = ([index] if return_index else []) + ([inverse] if return_inverse else [])
    return [vox] + extras if extras else vox


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
    """Samples -> one SparseTensor: rows concatenated in sample order, the sample number as a 4th coordinate column
    (TS:torchsparse/utils/collate.py:11-32). Both outputs are allocated once and filled slice by slice."""
    stride = inputs[0].stride
    rows = []
    for x in inputs:
        if isinstance(x.coords, np.ndarray):
            x.coords = torch.tensor(x.coords)
        if isinstance(x.feats, np.ndarray):
            x.feats = torch.tensor(x.feats)
        if not (isinstance(x.coords, torch.Tensor) and isinstance(x.feats, torch.Tensor)):
            raise AssertionError((type(x.coords), type(x.feats)))
        if x.stride != stride:
            raise AssertionError((x.stride, stride))
        rows.append(int(x.coords.shape[0]))
    first = inputs[0]
    total = sum(rows)
    width = first.coords.shape[1]
    coords = torch.empty((total, width + 1), dtype=torch.result_type(first.coords, torch.tensor(0, dtype=torch.int)),
                         device=first.coords.device)
    feats = torch.empty((total,) + tuple(first.feats.shape[1:]), dtype=first.feats.dtype, device=first.feats.device)
    lo = 0
    for b, (x, n) in enumerate(zip(inputs, rows)):
        coords[lo:lo + n, :width] = x.coords
        coords[lo:lo + n, width] = b
        feats[lo:lo + n] = x.feats
        lo += n
    return SparseTensor(coords=coords, feats=feats, stride=stride)


def sparse_collate_fn(inputs):
    """The dataloader's collate: a list of per-sample dicts -> one dict, by the type of each entry (SparseTensor: batched with a
    batch column; arrays / tensors: stacked; dicts: recursively; anything else: the list). TS:torchsparse/utils/collate.py:35-59."""
    if not isinstance(inputs[0], dict):
        return inputs

    def merge(values):
        head = values[0]
        if isinstance(head, SparseTensor):
            return sparse_collate(values)
        if isinstance(head, dict):
            return sparse_collate_fn(values)
        if isinstance(head, np.ndarray):
            return torch.stack([torch.tensor(v) for v in values], dim=0)
        if isinstance(head, torch.Tensor):
            return torch.stack(values, dim=0)
        return values

    return {name: merge([sample[name] for sample in inputs]) for name in inputs[0]}
