"""openpcseg_amd -- MI355X (gfx950) native sparse-voxel hot path behind the torchsparse 1.4
operator API that PJLab-ADG/OpenPCSeg's segmentors call.

    import openpcseg_amd
    openpcseg_amd.install_as_torchsparse()   # `import torchsparse` now resolves to this package
"""
from .sparse import SparseTensor, PointTensor, cat, fapply, get_kernel_offsets, make_ntuple  # noqa: F401
from .compat import install_as_torchsparse, install_reference_aliases  # noqa: F401

__version__ = "0.1.0"
