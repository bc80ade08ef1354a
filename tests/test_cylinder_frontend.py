"""Cylinder3D front-end: cart2polar + cylindrical partition + voxelize_with_label
(R:pcseg/data/dataset/semantickitti/semantickitti_cylinder.py:19-45,144-160) and the eval-time inverse-map argmax
(R:pcseg/model/segmentor/voxel/minkunet/minkunet.py:441-453).

Goldens (tests/golden/cylinder_golden.npz) come from RUNNING the reference's own SemkittiCylinderDataset.
get_single_sample on synthetic scans (`make_golden.py cylinder`). CPU: the oracle restatement is pinned to them,
bit-exact. `-m gpu`: the HIP kernels vs the goldens -- integers bit-exact, float features to 1e-6 -- with one
documented allowance: NumPy takes arctan2 from the host libm (atan2f, within 1 ulp on glibc < 2.41) and the kernel
rounds a double atan2 once, so a point whose angle sits within one float32 ulp of a cell face may fall into the
neighbouring cell; such points are counted, bounded (<= 1e-4 of the points) and each is verified to be a face case."""
import os

import numpy as np
import pytest
import torch

from oracle import oracle as orc

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cylinder_golden.npz")
CASES = ["cy480", "small", "clip"]


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


def _cfg(g, case):
    lo, hi, grid = g[case + "_cfg"]
    return lo.tolist(), hi.tolist(), grid.tolist()


@pytest.mark.parametrize("case", CASES)
def test_oracle_matches_reference_golden(gold, case):
    lo, hi, grid = _cfg(gold, case)
    pts, labels = gold[case + "_points"], gold[case + "_labels"]
    pol, coord, feat = orc.cylinder_partition(pts, lo, hi, grid)
    assert np.array_equal(coord.astype(np.float32), gold[case + "_point_coord"])
    assert np.array_equal(feat, gold[case + "_point_feature"])
    vox, vlab, inds, inv = orc.voxelize_with_label(coord, labels, 20)
    assert np.array_equal(vox, gold[case + "_voxel_coord"]) and np.array_equal(vlab, gold[case + "_voxel_label"])
    assert np.array_equal(inv, gold[case + "_inverse_map"])
    assert np.array_equal(feat[inds], gold[case + "_voxel_feature"])


def _face_case(pts, lo, hi, grid, coord_ref, coord_dev, rows):
    """Every row where the device cell differs from NumPy's must differ by one cell along phi only, with the angle
    within 2 float32 ulp of the face between the two cells."""
    pol = orc.cylinder_partition(pts, lo, hi, grid)[0]
    interval = (hi[1] - lo[1]) / (grid[1] - 1)
    for r in rows:
        d = coord_dev[r].astype(np.int64) - coord_ref[r].astype(np.int64)
        assert d[0] == 0 and d[2] == 0 and abs(d[1]) == 1, (r, d)
        face = lo[1] + max(coord_dev[r][1], coord_ref[r][1]) * interval
        assert abs(float(pol[r, 1]) - face) <= 2 * np.spacing(np.float32(abs(face))), (r, pol[r, 1], face)


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_device_front_end_matches_reference_golden(hip, gold, case):
    from openpcseg_amd import cylinder
    lo, hi, grid = _cfg(gold, case)
    pts, labels = gold[case + "_points"], gold[case + "_labels"]
    dpts = torch.from_numpy(pts).cuda()
    ret = cylinder.cylinder_sample(dpts, torch.from_numpy(labels).cuda(), lo, hi, grid, 20)
    coord = ret["point_coord"].cpu().numpy()
    ref_coord = gold[case + "_point_coord"]
    diff = np.nonzero((coord != ref_coord).any(axis=1))[0]
    assert diff.size <= max(1, int(1e-4 * pts.shape[0])), diff.size
    _face_case(pts, lo, hi, grid, ref_coord, coord, diff)
    feat = ret["point_feature"].cpu().numpy()
    same = np.setdiff1d(np.arange(pts.shape[0]), diff)
    ref_feat = gold[case + "_point_feature"]
    assert np.abs(feat[same] - ref_feat[same]).max() <= 1e-6 * max(1.0, np.abs(ref_feat).max())
    if diff.size == 0:  # the voxel set, its order, labels, representative rows and inverse map: bit-exact
        assert np.array_equal(ret["voxel_coord"].cpu().numpy(), gold[case + "_voxel_coord"])
        assert np.array_equal(ret["voxel_label"].cpu().numpy(), gold[case + "_voxel_label"])
        assert np.array_equal(ret["inverse_map"].cpu().numpy(), gold[case + "_inverse_map"])
        vf = ret["voxel_feature"].cpu().numpy()
        assert np.abs(vf - gold[case + "_voxel_feature"]).max() <= 1e-6 * max(1.0, np.abs(ref_feat).max())
    # the dedup + vote stage on the REFERENCE's cell indices: always bit-exact
    vox, vlab, inds, inv = cylinder.voxelize_with_label(torch.from_numpy(ref_coord.astype(np.int32)).cuda(),
                                                        torch.from_numpy(labels).cuda(), 20)
    assert np.array_equal(vox.cpu().numpy(), gold[case + "_voxel_coord"])
    assert np.array_equal(vlab.cpu().numpy(), gold[case + "_voxel_label"])
    assert np.array_equal(inv.cpu().numpy(), gold[case + "_inverse_map"])
    ovox, ovlab, oinds, oinv = orc.voxelize_with_label(ref_coord.astype(np.int32), labels, 20)
    assert np.array_equal(inds.cpu().numpy(), oinds)


@pytest.mark.gpu
def test_device_front_end_full_scan_vs_oracle(hip):
    """Full 120k-ray scan at the shipped cy480 grid (480 x 360 x 32): cells vs the NumPy oracle with the face-case
    allowance, and the size-independent properties (every point maps to the voxel holding its cell; labels are a
    majority of their voxel's points)."""
    from openpcseg_amd import cylinder
    from openpcseg_amd.workloads.synthetic import make_scan
    lo, hi, grid = [0, -180, -4], [50, 180, 2], [480, 360, 32]
    pts = make_scan(seed=2).astype(np.float32)
    rng = np.random.default_rng(3)
    labels = rng.integers(0, 20, size=pts.shape[0]).astype(np.int64)
    ret = cylinder.cylinder_sample(torch.from_numpy(pts).cuda(), torch.from_numpy(labels).cuda(), lo, hi, grid, 20)
    pol, ocoord, ofeat = orc.cylinder_partition(pts, lo, hi, grid)
    coord = ret["point_coord"].cpu().numpy()
    diff = np.nonzero((coord != ocoord).any(axis=1))[0]
    assert diff.size <= 12, diff.size
    _face_case(pts, lo, hi, grid, ocoord, coord, diff)
    vc, inv = ret["voxel_coord"].cpu().numpy(), ret["inverse_map"].cpu().numpy()
    assert np.array_equal(vc[inv], coord.astype(np.int64))
    ovox, ovlab, oinds, oinv = orc.voxelize_with_label(coord.astype(np.int32), labels, 20)
    assert np.array_equal(vc, ovox) and np.array_equal(ret["voxel_label"].cpu().numpy(), ovlab)
    assert np.array_equal(inv, oinv)
    with pytest.raises(IndexError):
        cylinder.voxelize_with_label(torch.from_numpy(coord.astype(np.int32)).cuda(),
                                     torch.full((pts.shape[0],), 25, dtype=torch.int64).cuda(), 20)


@pytest.mark.gpu
@pytest.mark.parametrize("c", [20, 19, 33, 64])
def test_inverse_map_argmax(hip, c):
    """`out[cur_scene_pts][cur_inv].argmax(1)[:num_points]` (minkunet.py:448-451) in one kernel."""
    from openpcseg_amd import cylinder
    g = torch.Generator(device="cuda").manual_seed(c)
    logits = torch.randn(5000, c, device="cuda", generator=g)
    inv = torch.randint(0, 5000, (37000,), device="cuda", generator=g)
    ref = logits[inv].argmax(1)
    assert torch.equal(cylinder.map_voxel_predictions(logits, inv), ref)
    assert torch.equal(cylinder.map_voxel_predictions(logits, inv, num_points=1234), ref[:1234])
    assert torch.equal(hip.rows_argmax_gather(logits), logits.argmax(1))
    # ties -> first maximum, like np.argmax
    t = torch.zeros(8, c, device="cuda")
    t[:, 5] = 1.0
    t[:, 11] = 1.0
    assert (hip.rows_argmax_gather(t) == 5).all()
