"""Cylinder3D front-end on the MI355X: the per-frame NumPy work of the reference's
cylinder dataset and its eval-time voxel -> point mapping, for scans that are already resident in HBM.

Same names, argument meaning and outputs as
  cart2polar            R:pcseg/data/dataset/semantickitti/semantickitti_cylinder.py:19-22
  voxelize_with_label   R:...semantickitti_cylinder.py:31-45
  cylinder_sample       the body of get_single_sample, R:...semantickitti_cylinder.py:144-173 (after augmentation)
  map_voxel_predictions R:pcseg/model/segmentor/voxel/minkunet/minkunet.py:441-453 (`out[scene][inv].argmax(1)`)
Device tensors in, device tensors out (pcs_cylinder_partition_f32, pcs_quantize_*, pcs_voxel_label_vote,
pcs_rows_argmax_gather_f32); there is no CPU path -- NumPy callers keep the reference's own dataset code.

Documented differences from the NumPy / torch originals (none reachable from the reference's datasets):
  * a point label outside [0, num_classes) other than 67 raises IndexError, also when it is NEGATIVE (NumPy would
    wrap-index the vote counter for -num_classes <= label < 0);
  * map_voxel_predictions skips NaN logits (an all-NaN row gives class 0; torch.argmax returns the NaN's index);
  * voxelize_with_label(check=True) reads the range flag back (one host sync per frame); check=False returns the
    device flag as a fifth value so a caller can test it once per batch.
"""
import torch

from . import native

IGNORE_VOTE_LABEL = 67  # semantickitti_cylinder.py:36


def cart2polar(input_xyz):
    """(n, >=3) float32 [x, y, z] on the device -> (n, 3) [rho, phi (radians), z]; float32 like NumPy on a float32 scan
    (arctan2 evaluated in double and rounded once, see csrc/cylinder.hip)."""
    x, y = input_xyz[:, 0], input_xyz[:, 1]
    rho = torch.sqrt(x * x + y * y)
    phi = torch.atan2(y.double(), x.double()).to(input_xyz.dtype)
    return torch.stack((rho, phi, input_xyz[:, 2]), dim=1)


def voxelize_with_label(point_coords, point_labels, num_classes, check=True):
    """-> (voxel_coords (m,3) int32, voxel_labels (m,) int64, inds (m,) int64, inverse_map (n,) int64): the reference's
    sparse_quantize(point_coords, return_index, return_inverse) + per-voxel majority label (first arg-max; labels
    equal to 67 are not counted)."""
    be = native.backend()
    vox, inds, inverse = be.quantize(point_coords, (1.0, 1.0, 1.0), True, True)
    labels, bad = be.voxel_label_vote(inverse, point_labels.reshape(-1).long(), vox.shape[0], num_classes,
                                      IGNORE_VOTE_LABEL)
    if not check:
        return vox, labels, inds, inverse, bad
    if int(bad.item()):
        raise IndexError("voxelize_with_label: a point label is outside [0, %d) (and is not %d)"
                         % (num_classes, IGNORE_VOTE_LABEL))
    return vox, labels, inds, inverse


def cylinder_sample(point, point_label, cylinder_space_min, cylinder_space_max, grid_size, num_classes):
    """point (n, >=4) float32 [x, y, z, intensity...] on the device (already augmented), point_label (n,) ->
    the dict get_single_sample returns (device tensors; 'point_coord' float32, the integer tensors int64)."""
    be = native.backend()
    polar, coord, point_feature = be.cylinder_partition(point, cylinder_space_min, cylinder_space_max, grid_size)
    voxel_coord, voxel_label, inds, inverse_map = voxelize_with_label(coord, point_label, num_classes)
    # voxel_feature = [voxel centre, polar[inds], xy[inds], extras[inds]]: the cell centre of a voxel IS the cell centre
    # of its representative point, so it is the representative's point_feature row (:155-156)
    voxel_feature = point_feature.index_select(0, inds)
    return {"point_feature": point_feature, "point_coord": coord.float(), "point_label": point_label.reshape(-1).long(),
            "voxel_feature": voxel_feature, "voxel_coord": voxel_coord.long(), "voxel_label": voxel_label,
            "inverse_map": inverse_map, "num_points": torch.tensor([point.shape[0]], device=point.device)}


def map_voxel_predictions(out, inverse_map, num_points=None):
    """Per-point class = argmax of the logits row of the point's voxel, without materialising out[inverse_map]
    (minkunet.py:448-451: `out[cur_scene_pts][cur_inv].argmax(1)[:num_points]`)."""
    pred = native.backend().rows_argmax_gather(out, inverse_map)
    return pred if num_points is None else pred[:int(num_points)]
