"""Deterministic synthetic LiDAR scans at SemanticKITTI shape.

64 rings x 1875 azimuth steps = 120 000 rays; scene = ground plane at z = -1.73 m plus a
closed wall of radius r(az) = clip(22 + sum_i 3 k_i sin(i az + phi_i), 6, 48); range noise
N(0, 0.02); intensity U(0,1). Then the dataset transform of the reference
(R:pcseg/data/dataset/semantickitti/semantickitti_voxel.py:112-120):
round(xyz / voxel) -> shift to >= 0 -> sparse_quantize. Seed 0 gives 92 321 voxels.
"""
import numpy as np
import torch

from ..hostdata import sparse_collate_fn, sparse_quantize, sparse_quantize_frames
from ..sparse import SparseTensor

N_RINGS, N_AZ = 64, 1875


def make_scan(seed=0, n_points=None):
    """(N,4) fp32 [x, y, z, intensity]; n_points subsamples (every k-th ray) for small cases."""
    rng = np.random.default_rng(seed)
    elev = np.deg2rad(np.linspace(-24.8, 2.0, N_RINGS))
    az = np.linspace(-np.pi, np.pi, N_AZ, endpoint=False)
    k = rng.normal(size=8)
    phi = rng.uniform(0, 2 * np.pi, size=8)
    wall = 22.0 + sum(3.0 * k[i] * np.sin((i + 1) * az + phi[i]) for i in range(8))
    wall = np.clip(wall, 6.0, 48.0)
    el, a = np.meshgrid(elev, az, indexing="ij")
    with np.errstate(divide="ignore"):
        ground = np.where(el < 0, 1.73 / np.tan(-el), np.inf)
    d = np.minimum(ground, wall[None, :]) + rng.normal(0, 0.02, size=el.shape)
    pts = np.stack([d * np.cos(a), d * np.sin(a), d * np.tan(el), rng.uniform(0, 1, size=el.shape)], axis=-1)
    pts = pts.reshape(-1, 4).astype(np.float32)
    if n_points is not None and n_points < pts.shape[0]:
        sel = np.linspace(0, pts.shape[0] - 1, n_points).astype(np.int64)
        pts = pts[sel]
    return pts


def voxelize_scan(points, voxel_size=0.05, num_classes=20, seed=0):
    """Reference dataset transform -> dict(lidar=SparseTensor, targets=SparseTensor) on the host."""
    pc = np.round(points[:, :3] / voxel_size).astype(np.int32)
    pc -= pc.min(0, keepdims=1)
    _, inds, inverse = sparse_quantize(pc, return_index=True, return_inverse=True)
    rng = np.random.default_rng(seed + 12345)
    labels = rng.integers(0, num_classes, size=points.shape[0]).astype(np.int64)
    lidar = SparseTensor(points[inds].astype(np.float32), pc[inds])
    targets = SparseTensor(labels[inds], pc[inds])
    return {"lidar": lidar, "targets": targets, "num_points": np.array([points.shape[0]])}


def make_batch(seeds, n_points=None, voxel_size=0.05, num_classes=20):
    """sparse_collate_fn over one synthetic frame per seed (host tensors)."""
    frames = [voxelize_scan(make_scan(s, n_points), voxel_size, num_classes, s) for s in seeds]
    batch = sparse_collate_fn(frames)
    batch["offset"] = torch.cumsum(torch.tensor([f["lidar"].coords.shape[0] for f in frames]), 0).int()
    return batch


def make_raw_batch(seeds, n_points=None, num_classes=20):
    """The batch BEFORE the dataset transform: raw scans (sum n_i, 4) fp32, their frame ids and per-point labels (host tensors; the
    same scans and labels make_batch voxelises on the host)."""
    pts = [make_scan(s, n_points) for s in seeds]
    labels = [np.random.default_rng(s + 12345).integers(0, num_classes, size=p.shape[0]).astype(np.int64) for s, p in zip(seeds, pts)]
    frames = np.concatenate([np.full(p.shape[0], i, dtype=np.int32) for i, p in enumerate(pts)])
    return {"points": torch.from_numpy(np.concatenate(pts)), "frames": torch.from_numpy(frames),
            "labels": torch.from_numpy(np.concatenate(labels)), "num_frames": len(seeds),
            "offsets": [0] + np.cumsum([p.shape[0] for p in pts]).tolist()}   # host-side frame boundaries (known when the scans are uploaded)


def device_collate(raw, voxel_size=0.05):
    """Dataset transform + sparse_quantize + sparse_collate_fn of a raw batch resident in HBM, bit-exact
    with voxelize_scan / make_batch on the host: round(xyz / voxel) -> shift every frame to >= 0
    (R:pcseg/data/dataset/semantickitti/semantickitti_voxel.py:112-120) -> voxel dedup of all frames in one pass
    (hostdata.sparse_quantize_frames) -> gathered features / labels with the batch column."""
    pts, frames, labels, nf = raw["points"], raw["frames"], raw["labels"], raw["num_frames"]
    # a TENSOR divisor: torch turns `tensor / python_float` into a multiplication by the reciprocal on the device, which is not
    # NumPy's float32 division (the voxel a point falls into may differ by one)
    v = torch.full((), voxel_size, dtype=torch.float32, device=pts.device)
    pc = torch.round(pts[:, :3] / v).int()
    off = raw["offsets"]
    sizes = {off[i + 1] - off[i] for i in range(nf)}
    if len(sizes) == 1:   # equal-length scans: one reduction for the per-frame minimum
        n = sizes.pop()
        pc = (pc.view(nf, n, 3) - pc.view(nf, n, 3).amin(1, keepdim=True)).view(-1, 3)
    else:
        pc = torch.cat([pc[off[i]:off[i + 1]] - pc[off[i]:off[i + 1]].amin(0, keepdim=True) for i in range(nf)])
    vox, index, _ = sparse_quantize_frames(pc, frames, nf)
    return {"lidar": SparseTensor(pts[index], vox), "targets": SparseTensor(labels[index], vox)}


class DeviceInputPrefetcher:
    """The input side of a training loop with the dataset transform on the device: `device_collate` of batch i + 1 runs on a SIDE
    stream while step i computes, like the reference's DataLoader workers prepare the next batch during the step
    (R:pcseg/data/__init__.py:106-124). `device_collate` reads one number back (the voxel count sizes the outputs); inside the
    step that read waits for everything queued before it and leaves the launch queue empty behind it -- 5 ms of a 55 ms step on
    the driver's box in round 5. Here the read waits for the side stream only, while the main stream holds a whole queued step.

        pipe = DeviceInputPrefetcher(lambda i: raw_batch_of(i))
        for i in range(steps):
            batch = pipe.next()          # waits (on the device) for the collate launched during the previous step
            ... forward / backward / optimizer step enqueued ...
            pipe.prefetch()              # collate of the next batch: side stream, host read hidden behind the queued step
    """

    def __init__(self, raw_of, voxel_size=0.05):
        self.raw_of, self.voxel_size = raw_of, voxel_size
        self.stream = torch.cuda.Stream()
        self.i = 0
        self._pending = None

    def _launch(self):
        raw = self.raw_of(self.i)
        self.i += 1
        self.stream.wait_stream(torch.cuda.current_stream())   # the raw scans may have been produced on the caller's stream
        with torch.cuda.stream(self.stream):
            batch = device_collate(raw, self.voxel_size)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        self._pending = (batch, ev)

    def prefetch(self):
        if self._pending is None:
            self._launch()

    def next(self):
        if self._pending is None:
            self._launch()
        batch, ev = self._pending
        self._pending = None
        cur = torch.cuda.current_stream()
        cur.wait_event(ev)
        for st in batch.values():          # allocated on the side stream, consumed on this one
            for t in (st.feats, st.coords):
                t.record_stream(cur)
        return batch
