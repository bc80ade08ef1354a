"""openpcseg_amd -- MI355X (gfx950) native sparse-voxel hot path behind the torchsparse 1.4
operator API that PJLab-ADG/OpenPCSeg's segmentors call.

    import openpcseg_amd
    openpcseg_amd.install_as_torchsparse()   # `import torchsparse` now resolves to this package
    openpcseg_amd.fuse(model)                # optional: block fusion for the reference's unmodified segmentors (block_fusion.py)
"""
from .sparse import SparseTensor, PointTensor, cat, fapply, get_kernel_offsets, make_ntuple  # noqa: F401
from .compat import install_as_torchsparse, install_reference_aliases  # noqa: F401


def fuse(model, criterion=True, glue=True, forward=True):
    """Block fusion for an unmodified reference segmentor (openpcseg_amd/block_fusion.py)."""
    from .block_fusion import fuse as _fuse
    return _fuse(model, criterion=criterion, glue=glue, forward=forward)


def unfuse(model):
    from .block_fusion import unfuse as _unfuse
    return _unfuse(model)

__version__ = "0.1.0"
