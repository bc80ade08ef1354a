"""Expose this package under the reference's import names.

`pcseg/model` and `tools/` import `torchsparse`, `torchsparse.nn as spnn`,
`torchsparse.nn.functional as F`, `torchsparse.nn.utils`, `torchsparse.utils.quantize`,
`torchsparse.utils.collate` (boundary B-A). install_as_torchsparse()
registers module objects with exactly those names in sys.modules so the segmentors load
unmodified.
"""
import sys
import types

TORCHSPARSE_VERSION = "1.4.0"


def _module(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    m.__all__ = [k for k in attrs if not k.startswith("_")]
    return m


def install_as_torchsparse(force=False, fuse=False):
    """fuse=True: every model is passed through `openpcseg_amd.fuse` the first time it is called (block fusion for the
    reference's unmodified segmentors: conv-epilogue BatchNorm statistics, BN + residual + ReLU in one pass, concat written by
    the apply pass, device Lovasz-softmax; see block_fusion.py). State dicts are unchanged either way."""
    if fuse:
        from .block_fusion import install_auto_fuse
        install_auto_fuse()
    if "torchsparse" in sys.modules and not force:
        existing = sys.modules["torchsparse"]
        if getattr(existing, "__openpcseg_amd__", False):
            return existing
        raise RuntimeError("a different `torchsparse` is already imported")
    from . import backend_shim
    from . import functional as Fn
    from . import hostdata, modules, sparse

    f_names = ["sphash", "sphashquery", "spcount", "spvoxelize", "spdevoxelize", "calc_ti_weights",
               "spdownsample", "conv3d", "relu", "leaky_relu"]
    functional = _module("torchsparse.nn.functional", **{n: getattr(Fn, n) for n in f_names})
    nn_utils = _module("torchsparse.nn.utils", fapply=sparse.fapply,
                       get_kernel_offsets=sparse.get_kernel_offsets)
    m_names = ["Conv3d", "BatchNorm", "ReLU", "LeakyReLU"]
    nn_mod = _module("torchsparse.nn", functional=functional, utils=nn_utils,
                     **{n: getattr(modules, n) for n in m_names})
    quantize = _module("torchsparse.utils.quantize", sparse_quantize=hostdata.sparse_quantize,
                       ravel_hash=hostdata.ravel_hash)
    collate = _module("torchsparse.utils.collate", sparse_collate=hostdata.sparse_collate,
                      sparse_collate_fn=hostdata.sparse_collate_fn)
    utils = _module("torchsparse.utils", make_ntuple=sparse.make_ntuple, quantize=quantize,
                    collate=collate)
    tensor = _module("torchsparse.tensor", SparseTensor=sparse.SparseTensor,
                     PointTensor=sparse.PointTensor)
    operators = _module("torchsparse.operators", cat=sparse.cat)
    b_names = [n for n in dir(backend_shim) if n.endswith("_cuda") or n.endswith("_cpu")]   # the 20 names of pybind_cuda.cpp:18-39
    backend = _module("torchsparse.backend", **{n: getattr(backend_shim, n) for n in b_names})
    top = _module("torchsparse", SparseTensor=sparse.SparseTensor, PointTensor=sparse.PointTensor,
                  cat=sparse.cat, nn=nn_mod, utils=utils, tensor=tensor, operators=operators, backend=backend,
                  __version__=TORCHSPARSE_VERSION)
    top.__openpcseg_amd__ = True
    top.__path__ = []  # mark as package so `import torchsparse.nn` resolves through sys.modules
    for m in (nn_mod, utils):
        m.__path__ = []
    for m in (top, nn_mod, functional, nn_utils, utils, quantize, collate, tensor, operators, backend):
        sys.modules[m.__name__] = m
    return top


def install_reference_aliases(fuse=False):
    """Everything the reference's sparse segmentors import from outside its own tree, served by this package:
    torchsparse (+ backend), torch_scatter (scatter_max / scatter_mean), range_utils (map_count / denselize)."""
    from .rangelib import install_as_range_utils
    from .scatter import install_as_torch_scatter
    install_as_torchsparse(fuse=fuse)
    install_as_torch_scatter()
    install_as_range_utils()
