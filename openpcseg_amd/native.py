"""ctypes binding of libpcseg_hip.so (include/pcseg_hip.h) + the tensor-level backend.

PyTorch is plumbing here: it owns device memory (caching allocator) and the current HIP
stream; every compute call below is one or more `pcs_*` C-ABI calls on raw device pointers.
There is NO CPU path: a missing library or a non-HIP tensor raises.
"""
import ctypes
import os
from ctypes import c_char_p, c_double, c_float, c_int32, c_int64, c_size_t, c_void_p

import torch

_LIB_PATH = os.environ.get("PCS_LIB_PATH") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib",
                                                          "libpcseg_hip.so")  # PCS_LIB_PATH: debug builds only

# symbol -> (restype, argtypes); kept in one table so tests can check every symbol that
# include/pcseg_hip.h declares is exported.
_P = c_void_p
SIGNATURES = {
    "pcs_abi_version": (c_int32, []),
    "pcs_last_error": (c_char_p, []),
    "pcs_hash": (c_int32, [_P, c_int64, _P, _P]),
    "pcs_kernel_hash": (c_int32, [_P, c_int64, _P, c_int32, _P, _P]),
    "pcs_hashtable_capacity": (c_int64, [c_int64]),
    "pcs_hashtable_bytes": (c_size_t, [c_int64]),
    "pcs_hashtable_build": (c_int32, [_P, c_int64, _P, c_int64, _P]),
    "pcs_hashtable_query": (c_int32, [_P, c_int64, _P, c_int64, _P, _P]),
    "pcs_count": (c_int32, [_P, c_int64, _P, c_int64, _P]),
    "pcs_voxelize_fwd_f32": (c_int32, [_P, _P, _P, c_int64, c_int64, c_int32, _P, _P]),
    "pcs_voxelize_fwd_csr_f32": (c_int32, [_P, _P, _P, _P, c_int64, c_int32, _P, _P]),
    "pcs_voxelize_bwd_f32": (c_int32, [_P, _P, _P, c_int64, c_int32, _P, _P]),
    "pcs_devoxelize_fwd_f32": (c_int32, [_P, _P, _P, c_int64, c_int32, _P, _P]),
    "pcs_devoxelize_bwd_f32": (c_int32, [_P, _P, _P, c_int64, c_int64, c_int32, _P, _P]),
    "pcs_devoxelize_bwd_csr_f32": (c_int32, [_P, _P, _P, _P, c_int64, c_int32, _P, _P]),
    "pcs_corner_map_f32": (c_int32, [_P, c_int32, c_int64, c_int32, _P, c_int64, _P, _P, _P]),
    "pcs_ti_weights_f32": (c_int32, [_P, c_int32, _P, c_int64, c_float, _P, _P]),
    "pcs_downsample_pack": (c_int32, [_P, c_int64, _P, c_int32, _P, c_int32, _P, _P, _P, _P]),
    "pcs_downsample_unpack": (c_int32, [_P, c_int64, _P, _P]),
    "pcs_index_csr_ws_bytes": (c_size_t, [c_int64, c_int64]),
    "pcs_index_csr_i32": (c_int32, [_P, c_int64, c_int64, _P, _P, _P, c_size_t, _P]),
    "pcs_sort_unique_ws_bytes": (c_size_t, [c_int64]),
    "pcs_sort_unique_i64": (c_int32, [_P, c_int64, _P, _P, _P, _P, c_size_t, _P]),
    "pcs_rulebook_ws_bytes": (c_size_t, [c_int64, c_int32]),
    "pcs_rulebook_probe": (c_int32, [_P, c_int64, _P, c_int32, _P, c_int64, _P, _P, _P, c_size_t, c_int32, _P]),
    "pcs_rulebook_fill": (c_int32, [_P, c_int64, c_int32, _P, _P, _P, _P]),
    "pcs_rulebook_tile_segments": (c_int32, [_P, _P, c_int32, c_int64, c_int32, c_int32, _P, _P]),
    "pcs_rulebook_tile_order": (c_int32, [_P, c_int32, c_int64, _P, _P]),
    "pcs_conv_kernel_revision": (c_char_p, []),
    "pcs_conv_tile_rows": (c_int32, [c_int32, c_int32]),
    "pcs_conv_pick_tile_rows": (c_int32, [c_int64, c_int64, c_int32, c_int32, c_int32]),
    "pcs_conv_pick_tile_rows_dt": (c_int32, [c_int64, c_int64, c_int32, c_int32, c_int32, c_int32]),
    "pcs_conv_emits_bn_partials": (c_int32, [c_int32, c_int32, c_int32, c_int32, c_int32]),
    "pcs_conv_uses_tile_order": (c_int32, [c_int32, c_int32, c_int32, c_int32]),
    "pcs_conv_gather_gemm_f32": (c_int32, [_P, c_int64, c_int32, _P, c_int32, c_int32, _P, c_int32,
                                           _P, c_int32, c_int64, _P, _P, _P, _P, _P]),
    "pcs_conv_gather_gemm_f32_ex": (c_int32, [_P, c_int64, c_int32, _P, c_int32, c_int32, _P, c_int32,
                                              _P, c_int32, c_int64, _P, _P, _P, _P, _P, _P]),
    "pcs_conv_supports_epilogue": (c_int32, [c_int32, c_int32, c_int32, c_int32]),
    "pcs_bn_bwd_reduce_partials": (c_int32, [_P, c_int64, c_int32, _P, c_int64, _P]),
    "pcs_bn_bwd_apply_act": (c_int32, [_P, _P, _P, _P, _P, _P, c_double, _P, _P, c_int64, c_int32, c_int32, c_int32, c_float, _P, _P,
                                       c_int64, _P]),
    "pcs_bn_reduce_partials": (c_int32, [_P, c_int64, c_int32, c_int64, _P, _P]),
    "pcs_bn_reduce_partials_finalize": (c_int32, [_P, c_int64, c_int32, c_int64, c_double, c_double, _P, _P, _P, _P, _P]),
    "pcs_transpose_kab_f32": (c_int32, [_P, c_int32, c_int32, c_int32, _P, _P]),
    "pcs_conv_wgrad_ws_bytes": (c_size_t, [_P, c_int32, c_int32, c_int32]),
    "pcs_conv_wgrad_f32": (c_int32, [_P, c_int32, _P, c_int32, _P, c_int32, _P, _P, c_int32, _P,
                                     _P, c_size_t, _P]),
    "pcs_scatter_max_fwd_f32": (c_int32, [_P, _P, _P, c_int64, c_int32, _P, _P, _P]),
    "pcs_scatter_max_bwd_f32": (c_int32, [_P, _P, c_int64, c_int64, c_int32, _P, _P]),
    "pcs_map_count": (c_int32, [_P, c_int64, c_int32, c_int32, c_int32, _P, _P]),
    "pcs_denselize_fwd_f32": (c_int32, [_P, _P, _P, c_int64, c_int32, c_int32, c_int32, c_int32, _P, _P]),
    "pcs_denselize_fwd_csr_f32": (c_int32, [_P, _P, _P, _P, c_int32, c_int32, c_int32, c_int32, _P, _P]),
    "pcs_denselize_bwd_csr_f32": (c_int32, [_P, _P, _P, _P, c_int64, c_int32, c_int32, c_int32, c_int32, _P, _P]),
    "pcs_denselize_bwd_f32": (c_int32, [_P, _P, _P, c_int64, c_int32, c_int32, c_int32, c_int32, _P, _P]),
    "pcs_range_sample_fwd_f32": (c_int32, [_P, _P, c_int64, c_int32, c_int32, c_int32, c_int32, _P, _P]),
    "pcs_range_sample_corners": (c_int32, [_P, c_int64, c_int32, c_int32, c_int32, _P, _P, _P]),
    "pcs_range_sample_bwd_csr_f32": (c_int32, [_P, _P, _P, _P, c_int32, c_int32, c_int32, c_int32, _P, _P]),
    "pcs_bn_num_partials": (c_int32, []),
    "pcs_bn_stats_f32": (c_int32, [_P, c_int64, c_int32, _P, _P, _P]),
    "pcs_bn_finalize_f32": (c_int32, [_P, c_double, _P, c_int32, c_double, c_double, _P, _P, _P, _P]),
    "pcs_bn_apply_f32": (c_int32, [_P, _P, _P, _P, _P, c_int64, c_int32, c_int32, _P, _P, c_int64, _P, c_int32, _P]),
    "pcs_bn_bwd_stats_f32": (c_int32, [_P, _P, _P, _P, _P, c_int64, c_int32, c_int32, _P, _P, c_int64, c_int64, _P]),
    "pcs_bn_bwd_apply_f32": (c_int32, [_P, _P, _P, _P, _P, _P, c_double, _P, _P, c_int64, c_int32, c_int32, _P, _P,
                                       c_int64, _P]),
    "pcs_bn_stats_h": (c_int32, [_P, c_int64, c_int32, c_int32, _P, _P, _P]),
    "pcs_bn_apply_h": (c_int32, [_P, _P, _P, _P, _P, c_int64, c_int32, c_int32, c_int32, _P, _P, c_int64, _P, c_int32, _P]),
    "pcs_bn_bwd_stats_h": (c_int32, [_P, _P, _P, _P, _P, c_int64, c_int32, c_int32, c_int32, _P, _P, c_int64, c_int64, _P]),
    "pcs_bn_bwd_apply_h": (c_int32, [_P, _P, _P, _P, _P, _P, c_double, _P, _P, c_int64, c_int32, c_int32, c_int32, _P, _P,
                                     c_int64, _P]),
    "pcs_quantize_floor": (c_int32, [_P, c_int32, c_int64, c_int32, _P, _P, _P, _P]),
    "pcs_quantize_keys": (c_int32, [_P, c_int64, _P, _P, _P]),
    "pcs_quantize_frame_keys": (c_int32, [_P, _P, c_int64, _P, _P, _P]),
    "pcs_quantize_flags": (c_int32, [_P, c_int64, _P, _P]),
    "pcs_quantize_emit": (c_int32, [_P, _P, _P, _P, c_int64, _P, _P, _P, _P]),
    "pcs_conv_h_applies": (c_int32, [c_int32, c_int32, c_int32]),
    "pcs_conv_prepared_weights_bytes": (c_size_t, [c_int32, c_int32, c_int32]),
    "pcs_conv_prepare_weights_h": (c_int32, [_P, c_int32, c_int32, c_int32, c_int32, c_int32, _P, _P]),
    "pcs_conv_gather_gemm_h": (c_int32, [_P, c_int64, c_int32, _P, c_int32, c_int32, _P, c_int32, _P, c_int32, c_int64,
                                         _P, _P, c_int32, _P, _P, _P]),
    "pcs_conv_gather_gemm_h_ex": (c_int32, [_P, c_int64, c_int32, _P, c_int32, c_int32, _P, c_int32, _P, c_int32, c_int64,
                                            _P, _P, _P, c_int32, _P, _P, _P]),
    "pcs_conv_x3_applies": (c_int32, [c_int32, c_int32, c_int32]),
    "pcs_conv_x3_column_tiles": (c_int32, [c_int32]),
    "pcs_conv_x3_emits_bn_partials": (c_int32, [c_int32, c_int32, c_int32, c_int32]),
    "pcs_conv_prepared_weights_x3_bytes": (c_size_t, [c_int32, c_int32, c_int32]),
    "pcs_conv_prepare_weights_x3": (c_int32, [_P, c_int32, c_int32, c_int32, c_int32, _P, _P]),
    "pcs_conv_gather_gemm_f32_bf16x3": (c_int32, [_P, c_int64, c_int32, _P, c_int32, c_int32, _P, c_int32, _P, c_int32, c_int64,
                                                  _P, _P, _P, _P, _P]),
    "pcs_conv_wgrad_f32_bf16x3": (c_int32, [_P, c_int32, _P, c_int32, _P, c_int32, _P, _P, c_int32, _P,
                                            _P, c_size_t, _P]),
    "pcs_conv_wgrad_h": (c_int32, [_P, c_int32, _P, c_int32, _P, c_int32, _P, _P, c_int32, _P, _P, c_size_t, c_int32, _P]),
    "pcs_unique_emit": (c_int32, [_P, _P, _P, _P, c_int64, _P, _P, _P, _P]),
    "pcs_cylinder_partition_f32": (c_int32, [_P, c_int64, c_int32, _P, _P, _P, _P, _P, _P, _P]),
    "pcs_voxel_label_vote": (c_int32, [_P, _P, c_int64, c_int64, c_int32, c_int64, _P, _P, _P, _P]),
    "pcs_rows_argmax_gather_f32": (c_int32, [_P, c_int64, c_int32, _P, c_int64, _P, _P]),
    "pcs_weights_multi_plan": (c_int64, [_P, c_int32]),
    "pcs_weights_multi": (c_int32, [_P, c_int32, c_int64, _P]),
    "pcs_lovasz_workspace_bytes": (c_int64, [c_int64, c_int32, c_int32, c_int64]),
    "pcs_lovasz_softmax_f32": (c_int32, [_P, _P, c_int64, c_int32, c_int32, c_int64, _P, _P, _P, c_int64, _P]),
    "pcs_debug_convh_ws": (None, [c_int32, c_int32, c_int32]),       # measurement switches (A/B tools), not part of the contract
    "pcs_debug_wgrad_interleave": (None, [c_int32]),
}

class _WeightJob(ctypes.Structure):   # pcs_weight_job of include/pcseg_hip.h
    _fields_ = [("src", c_void_p), ("dst", c_void_p), ("K", c_int32), ("A", c_int32), ("B", c_int32), ("kind", c_int32),
                ("transpose", c_int32), ("nctt", c_int32), ("nt16", c_int32), ("ns", c_int32), ("first_block", c_int64)]


ABI_VERSION = 11  # include/pcseg_hip.h PCS_ABI_VERSION (11: pcs_sort_unique_i64, pcs_index_csr_i32; 10: pcs_conv_gather_gemm_*_ex + pcs_conv_epilogue, pcs_bn_bwd_reduce_partials; 9: pcs_quantize_frame_keys; 8: sums2 size argument of pcs_bn_bwd_stats_*, ring switch removed; 7: pcs_lovasz_*; 6: ring kernel switch / query; 5: fp32 convolution on the bf16 MFMAs, pcs_conv_*_x3; 4: tile order)
_lib = None


def lib_path():
    return _LIB_PATH


def load_library():
    """dlopen the C-ABI library and declare every signature. Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise RuntimeError(
            "openpcseg_amd: %s is missing -- build it with `python -m openpcseg_amd.build` "
            "(hipcc, gfx950). There is no CPU fallback." % _LIB_PATH)
    lib = ctypes.CDLL(_LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError = symbol missing = broken build
        fn.restype = res
        fn.argtypes = args
    if lib.pcs_abi_version() != ABI_VERSION:
        raise RuntimeError("openpcseg_amd: ABI version mismatch")
    _lib = lib
    return lib


def _check(rc, what):
    if rc != 0:
        msg = load_library().pcs_last_error()
        raise RuntimeError("openpcseg_amd: %s failed (%d): %s" % (what, rc, msg.decode() if msg else ""))


def _dev(t, name, dtype=None):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError("openpcseg_amd: `%s` must be a HIP device tensor (got %s); the MI355X "
                           "backend has no CPU path" % (name, getattr(t, "device", type(t))))
    if dtype is not None and t.dtype != dtype:
        raise TypeError("openpcseg_amd: `%s` must be %s, got %s" % (name, dtype, t.dtype))
    return t if t.is_contiguous() else t.contiguous()


def _h2d(values, dtype, device):
    """Small host list -> device tensor without stalling the host: staged in pinned memory, copied asynchronously on
    the current stream (a pageable-memory copy waits for everything already queued on the GPU)."""
    return torch.tensor(values, dtype=dtype, pin_memory=True).to(device, non_blocking=True)


def _ptr(t):
    return c_void_p(t.data_ptr()) if t is not None and t.numel() > 0 else c_void_p(0)


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
    """The current HIP stream of the current device as a raw handle. The raw-handle getter is ~20x cheaper than
    building a torch.cuda.Stream object, and a training step makes ~1000 of these calls."""
    if _raw_stream is not None and _cur_device is not None:
        return c_void_p(_raw_stream(_cur_device()))
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def _cache_key(t):
    """Identity of a tensor's CONTENT for the caches hung on caller tensors: storage address, shape and the in-place
    version counter. An in-place update (`copy_`, `+=`, index assignment) bumps `_version`, so a static input buffer
    refilled between steps never sees a stale table / CSR (the reference recomputes hash, query and sort on every
    call and has no such hazard). Tensors made under torch.inference_mode() get a unique key (recomputed each call)."""
    try:
        version = t._version
    except RuntimeError:  # inference tensors track no version counter: never served from a cache
        _cache_key.uncached += 1
        return ("uncached", _cache_key.uncached)
    return (t.data_ptr(), tuple(t.shape), version)


_cache_key.uncached = 0


# Memo bookkeeping for measurement tools: `epoch` is bumped by the caller once per step (tools/modelbench.py); a hit on a memo made
# in an earlier epoch (`cross`) means a step reused work of a previous step through a tensor object that outlived it.
CACHE_STATS = {"hit": 0, "miss": 0, "cross": 0, "epoch": 0}


def _cached(holder, name, key, make):
    """holder.<name> = (key, value[, epoch]) memo; rebuilt when the key differs. Tensors that refuse attributes just recompute."""
    hit = getattr(holder, name, None)
    if hit is not None and hit[0] == key:
        CACHE_STATS["hit"] += 1
        if len(hit) > 2 and hit[2] != CACHE_STATS["epoch"]:
            CACHE_STATS["cross"] += 1
        return hit[1]
    CACHE_STATS["miss"] += 1
    value = make()
    try:
        setattr(holder, name, (key, value, CACHE_STATS["epoch"]))
    except AttributeError:
        pass
    return value


class HashTable:
    """Device open-addressing table over a vector of 60-bit hashes (value = position)."""

    __slots__ = ("storage", "capacity", "n")

    def __init__(self, storage, capacity, n):
        self.storage, self.capacity, self.n = storage, capacity, n


_PPR_HINT = {}  # pairs per destination row last seen for a map family (kernel / stride key): launch-shape estimates


class KernelMap:
    """A rulebook: pairs (P,2) int32 = (src_row, dst_row), k-major, dst ascending within k.

    `koff` are the K+1 slice offsets (device int32) and `koff_host` the same numbers on the host; `nbsizes` (K,) int64
    mirrors the reference's kmap entry (TS:torchsparse/nn/functional/conv.py:168). Tile-segment tables are cached per
    tile height.

    Deferred sizes: a map built by `HipBackend.build_kmap` does not wait for its per-offset sizes. The pair list lives
    in a buffer sized for the worst case (K * n_dst rows), the sizes travel to pinned host memory asynchronously, and
    `koff_host` / `num_pairs` / `pairs` block on that copy only when first read -- in practice in the backward pass
    (weight-gradient plan), long after it has landed. The forward launches (segment tables, fused conv) only need the
    device-side `koff`, so a forward pass no longer drains the GPU queue once per map.
    """

    def __init__(self, pairs, koff, koff_host, nbsizes, n_src, n_dst, pending=None, hint_key=None):
        self._pairs_raw, self.koff, self.nbsizes = pairs, koff, nbsizes
        self.n_src, self.n_dst = n_src, n_dst
        self._koff_host, self._pending, self.hint_key = koff_host, pending, hint_key
        self.K = int(nbsizes.shape[0]) if koff_host is None else len(koff_host) - 1
        self._seg = {}
        self._koff_c_cache = None

    def _resolve(self):
        if self._koff_host is None:
            sizes, event = self._pending
            event.synchronize()
            ko = [0]
            for v in sizes.tolist():
                ko.append(ko[-1] + int(v))
            self._koff_host, self._pending = ko, None
            # exact-size copy: the worst-case buffer (K * n_dst rows, ~5x the pairs of a stride-1 map) is released
            # instead of living through backward behind a view; segment tables built from it stay valid (same rows)
            self._pairs_raw = self._pairs_raw[:ko[-1]].clone()
            if self.hint_key is not None and self.n_dst > 0:
                _PPR_HINT[self.hint_key] = ko[-1] / float(self.n_dst)
        return self._koff_host

    @property
    def resolved(self):
        return self._koff_host is not None

    @property
    def koff_host(self):
        return self._resolve()

    @property
    def pairs(self):
        self._resolve()
        return self._pairs_raw

    @property
    def _koff_c(self):
        if self._koff_c_cache is None:
            self._koff_c_cache = (c_int32 * (self.K + 1))(*self._resolve())
        return self._koff_c_cache

    @property
    def num_pairs(self):
        return self._resolve()[-1]

    def num_pairs_estimate(self):
        """Exact once the sizes are on the host; before that, the last pairs-per-row seen for this map family
        (only the launch shape of the fused conv depends on it)."""
        if self._koff_host is not None:
            return self._koff_host[-1]
        return int(_PPR_HINT.get(self.hint_key, 0.22 * self.K) * self.n_dst)

    def mirror(self):
        """The input-sorted map of a submanifold convolution with point-symmetric offsets (off[K-1-k] == -off[k]),
        without probing: its offset-k slice IS the offset-(K-1-k) slice of this map (pairs (r, q) with
        coord[r] = coord[q] - off[k], q ascending, r and q rows of the same coordinate set)."""
        k, ko = self.K, self.koff_host
        pairs = torch.cat([self._pairs_raw[ko[k - 1 - j]:ko[k - j]] for j in range(k)], dim=0)
        koff_host = [0]
        for j in range(k):
            koff_host.append(koff_host[-1] + ko[k - j] - ko[k - 1 - j])
        koff = _h2d(koff_host, torch.int32, self._pairs_raw.device)
        return KernelMap(pairs, koff, koff_host, self.nbsizes.flip(0), self.n_dst, self.n_src)


class HipBackend:
    """Tensor-level view of the C ABI (one method per reference backend function / fused op)."""

    name = "hip-gfx950"

    def __init__(self):
        self.lib = load_library()

    # -- K1 / K2 ------------------------------------------------------------------------------
    def hash(self, coords):
        coords = _dev(coords, "coords", torch.int32)
        out = torch.empty(coords.shape[0], dtype=torch.int64, device=coords.device)
        _check(self.lib.pcs_hash(_ptr(coords), coords.shape[0], _ptr(out), _stream()), "pcs_hash")
        return out

    def kernel_hash(self, coords, offsets):
        coords = _dev(coords, "coords", torch.int32)
        offsets = _dev(offsets, "offsets", torch.int32)
        n, k = coords.shape[0], offsets.shape[0]
        out = torch.empty((k, n), dtype=torch.int64, device=coords.device)
        _check(self.lib.pcs_kernel_hash(_ptr(coords), n, _ptr(offsets), k, _ptr(out), _stream()),
               "pcs_kernel_hash")
        return out

    # -- K3-K5 ---------------------------------------------------------------------------------
    def table_build(self, keys):
        keys = _dev(keys, "references", torch.int64)
        n = keys.numel()
        cap = self.lib.pcs_hashtable_capacity(n)
        storage = torch.empty(self.lib.pcs_hashtable_bytes(cap), dtype=torch.uint8, device=keys.device)
        _check(self.lib.pcs_hashtable_build(_ptr(keys), n, _ptr(storage), cap, _stream()),
               "pcs_hashtable_build")
        return HashTable(storage, cap, n)

    def table_query(self, table, queries):
        """-> int64, position + 1 or 0 (the reference backend's convention)."""
        queries = _dev(queries, "queries", torch.int64)
        out = torch.empty(queries.numel(), dtype=torch.int64, device=queries.device)
        _check(self.lib.pcs_hashtable_query(_ptr(table.storage), table.capacity, _ptr(queries),
                                            queries.numel(), _ptr(out), _stream()), "pcs_hashtable_query")
        return out

    def hash_query(self, queries, references):
        """sphashquery core: index of each query hash among references, -1 when absent."""
        table = self.table_build(references)
        return self.table_query(table, queries.reshape(-1)) - 1

    # -- K6 ------------------------------------------------------------------------------------
    def count(self, idx, num):
        idx = _dev(idx, "coords", torch.int32)
        out = torch.empty(int(num), dtype=torch.int32, device=idx.device)
        _check(self.lib.pcs_count(_ptr(idx), idx.numel(), _ptr(out), int(num), _stream()), "pcs_count")
        return out

    # -- K7-K10 ---------------------------------------------------------------------------------
    def voxelize_fwd(self, feats, idx, counts, cache_on=None):
        """out[v] = sum over the points i of voxel v of feats[i] / counts[v]: segmented over the points sorted by
        voxel (no atomics, deterministic). The sorted order is cached on `cache_on` (the caller's index tensor,
        which point_to_voxel reuses at every stage) or on idx."""
        feats = _dev(feats, "feats", torch.float32)
        idx = _dev(idx, "coords", torch.int32)
        counts = _dev(counts, "counts", torch.int32)
        c = feats.shape[1]
        m = counts.shape[0]
        holder = cache_on if cache_on is not None else idx
        csr = _cached(holder, "_pcs_vox_csr", _cache_key(holder) + (m, idx.shape[0]), lambda: self._csr(idx, m))
        out = torch.empty((m, c), dtype=torch.float32, device=feats.device)
        _check(self.lib.pcs_voxelize_fwd_csr_f32(_ptr(feats), _ptr(csr[0]), _ptr(csr[1]), _ptr(counts), m, c,
                                                 _ptr(out), _stream()), "pcs_voxelize_fwd_csr_f32")
        return out

    def voxelize_fwd_atomic(self, feats, idx, counts):
        """The reference's literal K7 dataflow (fp32 atomics); kept for A/B measurements."""
        feats = _dev(feats, "feats", torch.float32)
        idx = _dev(idx, "coords", torch.int32)
        counts = _dev(counts, "counts", torch.int32)
        n, c = feats.shape
        m = counts.shape[0]
        out = torch.empty((m, c), dtype=torch.float32, device=feats.device)
        _check(self.lib.pcs_voxelize_fwd_f32(_ptr(feats), _ptr(idx), _ptr(counts), n, m, c, _ptr(out),
                                             _stream()), "pcs_voxelize_fwd_f32")
        return out

    def voxelize_bwd(self, gout, idx, counts, n):
        gout = _dev(gout, "grad_output", torch.float32)
        c = gout.shape[1]
        gin = torch.empty((n, c), dtype=torch.float32, device=gout.device)
        _check(self.lib.pcs_voxelize_bwd_f32(_ptr(gout), _ptr(idx), _ptr(counts), n, c, _ptr(gin),
                                             _stream()), "pcs_voxelize_bwd_f32")
        return gin

    def devoxelize_fwd(self, feats, idx8, w8):
        feats = _dev(feats, "feats", torch.float32)
        idx8 = _dev(idx8, "coords", torch.int32)
        w8 = _dev(w8, "weights", torch.float32)
        n, c = idx8.shape[0], feats.shape[1]
        out = torch.empty((n, c), dtype=torch.float32, device=feats.device)
        _check(self.lib.pcs_devoxelize_fwd_f32(_ptr(feats), _ptr(idx8), _ptr(w8), n, c, _ptr(out),
                                               _stream()), "pcs_devoxelize_fwd_f32")
        return out

    def devoxelize_bwd(self, gout, idx8, w8, m):
        """gfeat[v] = sum over (point i, corner k) with idx8[i,k] == v of w8[i,k] * gout[i].
        Contention-free: entries sorted by voxel once per idx8 tensor (cached on the tensor; the
        same map serves every backward of one forward), then a segmented reduction."""
        gout = _dev(gout, "grad_output", torch.float32)
        n, c = gout.shape
        csr = _cached(idx8, "_pcs_csr", _cache_key(idx8) + (m,), lambda: self._csr(idx8, m))
        gfeat = torch.empty((m, c), dtype=torch.float32, device=gout.device)
        _check(self.lib.pcs_devoxelize_bwd_csr_f32(_ptr(gout), _ptr(csr[0]), _ptr(csr[1]), _ptr(w8), m, c,
                                                   _ptr(gfeat), _stream()), "pcs_devoxelize_bwd_csr_f32")
        return gfeat

    def devoxelize_bwd_atomic(self, gout, idx8, w8, m):
        """The reference's literal K10 dataflow (fp32 atomics); kept for A/B measurements."""
        gout = _dev(gout, "grad_output", torch.float32)
        n, c = gout.shape
        gfeat = torch.empty((m, c), dtype=torch.float32, device=gout.device)
        _check(self.lib.pcs_devoxelize_bwd_f32(_ptr(gout), _ptr(idx8), _ptr(w8), n, m, c, _ptr(gfeat),
                                               _stream()), "pcs_devoxelize_bwd_f32")
        return gfeat

    def level_table(self, voxel_coords):
        """Hash table over one level's voxel coordinates, cached on the coordinate tensor (shared with its kernel maps)."""
        voxel_coords = _dev(voxel_coords, "coords", torch.int32)
        return _cached(voxel_coords, "_pcs_table", _cache_key(voxel_coords),
                       lambda: self.table_build(self.hash(voxel_coords)))

    def corner_map(self, point_coords, voxel_coords, stride):
        """(idx8 (N,8) int32, w8 (N,8) float32) of voxel_to_point in one kernel: rows of the 8 corner voxels of every
        point's stride-`stride` cell and the trilinear weights (R:.../minkunet/utils.py:69-105)."""
        pc = _dev(point_coords, "coords", torch.float32)
        table = self.level_table(voxel_coords)
        n = pc.shape[0]
        idx8 = torch.empty((n, 8), dtype=torch.int32, device=pc.device)
        w8 = torch.empty((n, 8), dtype=torch.float32, device=pc.device)
        _check(self.lib.pcs_corner_map_f32(_ptr(pc), pc.shape[1], n, int(stride), _ptr(table.storage), table.capacity,
                                           _ptr(idx8), _ptr(w8), _stream()), "pcs_corner_map_f32")
        return idx8, w8

    def ti_weights(self, coords, idx_query, scale):
        coords = _dev(coords, "coords", torch.float32)
        idx_query = _dev(idx_query, "idx_query", torch.int64)
        n = coords.shape[0]
        w = torch.empty((8, n), dtype=torch.float32, device=coords.device)
        _check(self.lib.pcs_ti_weights_f32(_ptr(coords), coords.shape[1], _ptr(idx_query), n,
                                           float(scale), _ptr(w), _stream()), "pcs_ti_weights_f32")
        return w

    # -- spdownsample ---------------------------------------------------------------------------
    def sort_unique(self, keys, err=None):
        """Distinct int64 keys in ascending order (pcs_sort_unique_i64). Returns (buf, info): buf holds the info[0] distinct keys
        at its front, info = device int64 [count, largest key, *err]; no host read here."""
        keys = _dev(keys, "keys", torch.int64)
        nk = keys.numel()
        buf = torch.empty(nk, dtype=torch.int64, device=keys.device)
        info = torch.empty(3, dtype=torch.int64, device=keys.device)
        ws_bytes = self.lib.pcs_sort_unique_ws_bytes(nk)
        ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=keys.device)
        _check(self.lib.pcs_sort_unique_i64(_ptr(keys), nk, _ptr(buf), _ptr(info), _ptr(err) if err is not None else None, _ptr(ws),
                                            ws_bytes, _stream()), "pcs_sort_unique_i64")
        return buf, info

    def downsample(self, coords, sample_stride, offsets=None, coords_min=None):
        """Unique, (b,x,y,z)-sorted output coordinates. offsets=None -> fast branch."""
        coords = _dev(coords, "coords", torch.int32)
        n = coords.shape[0]
        ss = (c_int32 * 3)(*[int(s) for s in sample_stride])
        err = torch.zeros(1, dtype=torch.int32, device=coords.device)
        if offsets is None:
            keys = torch.empty(n, dtype=torch.int64, device=coords.device)
            rc = self.lib.pcs_downsample_pack(_ptr(coords), n, ss, 0, None, 0, None, _ptr(keys),
                                              _ptr(err), _stream())
        else:
            offsets = _dev(offsets, "offsets", torch.int32)
            coords_min = _dev(coords_min, "coords_min", torch.int32)
            k = offsets.shape[0]
            keys = torch.empty(n * k, dtype=torch.int64, device=coords.device)
            rc = self.lib.pcs_downsample_pack(_ptr(coords), n, ss, 1, _ptr(offsets), k,
                                              _ptr(coords_min), _ptr(keys), _ptr(err), _stream())
        _check(rc, "pcs_downsample_pack")
        # sorted distinct keys == the reference's lexicographic (b,x,y,z) unique rows (pcs_sort_unique_i64). ONE host read per
        # call -- the count that sizes the output, the largest key (the general branch's "rejected candidate" sentinel sorts
        # last) and the range error flag in one pinned record (torch.unique + the flag read were two round trips)
        buf, info = self.sort_unique(keys, err)
        m, last, bad = info.tolist()
        if bad:
            raise RuntimeError("openpcseg_amd: spdownsample coordinate out of the packed range "
                               "(|x|,|y|,|z| < 2^17, 0 <= batch < 512)")
        if offsets is not None and m > 0 and last == 0x7FFFFFFFFFFFFFFF:
            m -= 1
        uniq = buf[:m]
        m = uniq.numel()
        out = torch.empty((m, 4), dtype=torch.int32, device=coords.device)
        _check(self.lib.pcs_downsample_unpack(_ptr(uniq), m, _ptr(out), _stream()), "pcs_downsample_unpack")
        return out

    # -- rulebook -------------------------------------------------------------------------------
    def build_kmap(self, ref_coords, query_coords, offsets, hint_key=None, symmetric=False):
        """pairs (ref_row, query_row) for hash(query + offsets[k]) == hash(ref); k-major, query
        ascending. No host sync: the per-offset sizes are copied to pinned memory asynchronously and read when
        `koff_host` / `pairs` / `num_pairs` of the map are first needed (KernelMap, deferred sizes). symmetric: the
        caller knows offsets[K-1-k] == -offsets[k]; with ref_coords is query_coords (submanifold map) only half of the
        offsets are probed and the other half mirrored."""
        ref_coords = _dev(ref_coords, "coords", torch.int32)
        query_coords = _dev(query_coords, "coords", torch.int32)
        offsets = _dev(offsets, "offsets", torch.int32)
        dev = ref_coords.device
        table = self.level_table(ref_coords)  # shared by the level's submanifold / strided / transposed maps
        nq, k = query_coords.shape[0], offsets.shape[0]
        results = torch.empty((k, max(nq, 1)), dtype=torch.int32, device=dev)
        nbsizes = torch.empty(k, dtype=torch.int64, device=dev)
        ws_bytes = self.lib.pcs_rulebook_ws_bytes(nq, k)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        _check(self.lib.pcs_rulebook_probe(_ptr(query_coords), nq, _ptr(offsets), k, _ptr(table.storage),
                                           table.capacity, _ptr(results), _ptr(nbsizes), _ptr(ws),
                                           ws_bytes, int(bool(symmetric) and ref_coords is query_coords and k % 2 == 1),
                                           _stream()), "pcs_rulebook_probe")
        sizes = torch.empty(k, dtype=torch.int64, pin_memory=True)
        sizes.copy_(nbsizes, non_blocking=True)
        event = torch.cuda.Event()
        event.record()
        pairs = torch.empty((k * nq, 2), dtype=torch.int32, device=dev)  # worst case: every offset hits for every row
        koff = torch.empty(k + 1, dtype=torch.int32, device=dev)
        _check(self.lib.pcs_rulebook_fill(_ptr(results), nq, k, _ptr(ws), _ptr(pairs), _ptr(koff),
                                          _stream()), "pcs_rulebook_fill")
        return KernelMap(pairs, koff, None, nbsizes, ref_coords.shape[0], nq, pending=(sizes, event), hint_key=hint_key)

    def tile_rows(self, cin, cout, kmap=None, dtype=0):
        """Output tile height of one conv launch: the library default, or with a kernel map the per-layer pick
        (dtype 0: fp32 kernels, 1 / 2: bf16 / fp16 kernels)."""
        if kmap is None:
            return self.lib.pcs_conv_tile_rows(cin, cout)
        return self.lib.pcs_conv_pick_tile_rows_dt(kmap.n_dst, kmap.num_pairs_estimate(), kmap.K, cin, cout, dtype)

    def _segments(self, kmap, tile_rows):
        seg = kmap._seg.get(tile_rows)
        if seg is None:
            ntiles = (kmap.n_dst + tile_rows - 1) // tile_rows
            seg = torch.empty(kmap.K * (ntiles + 1), dtype=torch.int32, device=kmap._pairs_raw.device)
            _check(self.lib.pcs_rulebook_tile_segments(_ptr(kmap._pairs_raw), _ptr(kmap.koff), kmap.K,
                                                       kmap.n_dst, tile_rows, 1, _ptr(seg), _stream()),
                   "pcs_rulebook_tile_segments")
            kmap._seg[tile_rows] = seg
        return seg

    @staticmethod
    def _wants_order(kmap):
        """The heaviest-first order pays where workgroups are long and few and differ in work: the dense, coordinate-ordered
        levels (>= 6 pairs per row: +4..8 % per launch). The stride-1 level is in hash order inside the network (all tiles
        alike: +-1 %), the stride-2 level would gain 4 % for 2.3x the HBM reads of the XCD-contiguous row order
        (profiles/round6_conv_xcd_order_ab2.txt, round6_conv_order_traffic2.txt): those launches keep the row order."""
        return kmap.n_dst > 0 and kmap.num_pairs_estimate() >= 6.0 * kmap.n_dst

    def _tile_order(self, kmap, tile_rows):
        """Heaviest-first order of the row tiles of (kmap, tile_rows), cached beside the segment table it is made from
        (one small kernel per map and tile height, shared by every layer, forward and backward, that uses the map)."""
        key = ("order", tile_rows)
        order = kmap._seg.get(key)
        if order is None:
            seg = self._segments(kmap, tile_rows)
            ntiles = (kmap.n_dst + tile_rows - 1) // tile_rows
            order = torch.empty(max(ntiles, 1), dtype=torch.int32, device=seg.device)
            _check(self.lib.pcs_rulebook_tile_order(_ptr(seg), kmap.K, ntiles, _ptr(order), _stream()),
                   "pcs_rulebook_tile_order")
            kmap._seg[key] = order
        return order

    # -- convolution ----------------------------------------------------------------------------
    def _bn_partial(self, kmap, t, cin, cout, k, dtype_code, bn_sums, device):
        """Workspace for the BatchNorm partials of one conv launch, or None when not asked for / not produced."""
        if bn_sums is None or kmap.n_dst == 0:
            return None
        if not self.lib.pcs_conv_emits_bn_partials(cin, cout, k, t, dtype_code):
            return None
        ntiles = (kmap.n_dst + t - 1) // t
        return torch.empty(ntiles * 2 * cout, dtype=torch.float64, device=device)

    supports_bn_raw = True   # the conv entries take bn_raw=True: hand on the per-tile partials, not the reduced vector

    def _bn_reduce(self, partial, t, kmap, cout, bn_sums, raw=False):
        if raw:   # the consumer reduces (and, single process, finalizes in the same launch: bn_reduce_finalize)
            bn_sums.append(partial)
            return
        bn_sums.append(self.bn_reduce_partials(partial, cout, kmap.n_dst))

    def bn_reduce_partials(self, partial, c, n):
        """[ntiles][2][c] double partials of a conv write-back -> the (2c + 1) `sums` vector bn_stats would return."""
        partial = _dev(partial, "partial", torch.float64)
        sums = torch.empty(2 * c + 1, dtype=torch.float64, device=partial.device)
        _check(self.lib.pcs_bn_reduce_partials(_ptr(partial), partial.numel() // (2 * c), c, n, _ptr(sums), _stream()),
               "pcs_bn_reduce_partials")
        return sums

    def bn_reduce_finalize(self, partial, c, n, eps, momentum, running_mean, running_var):
        """bn_reduce_partials + bn_finalize(count = n) in one launch -> stat (2c); single-process forward only."""
        partial = _dev(partial, "partial", torch.float64)
        stat = torch.empty(2 * c, dtype=torch.float64, device=partial.device)
        _check(self.lib.pcs_bn_reduce_partials_finalize(_ptr(partial), partial.numel() // (2 * c), c, n, float(eps),
                                                        float(momentum),
                                                        _ptr(running_mean) if running_mean is not None else None,
                                                        _ptr(running_var) if running_var is not None else None, None,
                                                        _ptr(stat), _stream()), "pcs_bn_reduce_partials_finalize")
        return stat

    def conv_supports_addend(self, cin, cout, k, dtype=0):
        """The shape's kernel takes the write-back extras (addend, BatchNorm backward statistics: pcs_conv_epilogue)."""
        return bool(self.lib.pcs_conv_supports_epilogue(int(cin), int(cout), int(k), int(dtype)))

    def conv_emits_stats(self, cin, cout, k, kmap, half_dtype=None):
        """The default launch of this shape on this map leaves per-tile statistics in its write-back."""
        code = self._HALF[half_dtype] if half_dtype is not None else 0
        if kmap.n_dst == 0 or cout % 4:
            return False
        return bool(self.lib.pcs_conv_emits_bn_partials(cin, cout, k, self.tile_rows(cin, cout, kmap, code), code))

    class _Epilogue(ctypes.Structure):   # include/pcseg_hip.h: pcs_conv_epilogue
        _fields_ = [("addend", ctypes.c_void_p), ("bn_x", ctypes.c_void_p), ("bn_mask", ctypes.c_void_p), ("bn_stat", ctypes.c_void_p),
                    ("act_slope", ctypes.c_float), ("reserved", ctypes.c_int32)]

    def _epilogue(self, addend, bn_bwd, kmap, cout, dtype, t, cin, k, code, device, act_slope=None):
        """-> (ctypes pointer or None, keep-alive tuple, partial workspace or None) for the _ex entries.
        bn_bwd = (x, mask or None, stat): the BatchNorm whose output gradient this launch writes."""
        if addend is None and bn_bwd is None and act_slope is None:
            return None, None, None
        ep, part = self._Epilogue(None, None, None, None, float(act_slope) if act_slope is not None else 0.0, 0), None
        if addend is not None:
            addend = _dev(addend, "addend", dtype)
            if tuple(addend.shape) != (kmap.n_dst, cout):
                raise ValueError("addend %s does not match the output (%d, %d)" % (tuple(addend.shape), kmap.n_dst, cout))
            ep.addend = addend.data_ptr()
        if bn_bwd is not None:
            x, mask, stat = bn_bwd
            x = _dev(x, "bn_x", dtype)
            if tuple(x.shape) != (kmap.n_dst, cout):
                raise ValueError("bn_x %s does not match the output (%d, %d)" % (tuple(x.shape), kmap.n_dst, cout))
            if not self.lib.pcs_conv_emits_bn_partials(cin, cout, k, t, code):
                raise RuntimeError("openpcseg_amd: this shape / tile height leaves no statistics in its write-back")
            stat = _dev(stat, "bn_stat", torch.float64)
            ntiles = (kmap.n_dst + t - 1) // t
            part = torch.empty(ntiles * 2 * cout, dtype=torch.float64, device=device)
            ep.bn_x, ep.bn_stat = x.data_ptr(), stat.data_ptr()
            if mask is not None:
                ep.bn_mask = _dev(mask, "bn_mask").data_ptr()
            bn_bwd = (x, mask, stat)
        return ctypes.byref(ep), (ep, addend, bn_bwd), part

    def bn_bwd_reduce_partials(self, partial, c):
        """per-tile [2][c] double partials of a dgrad write-back -> the sums2 vector bn_bwd_stats returns (with its fp32 copy)."""
        partial = _dev(partial, "partial", torch.float64)
        buf = torch.empty(3 * c, dtype=torch.float64, device=partial.device)
        sums2 = buf[:2 * c]
        sums2._pcs_f32 = buf[2 * c:].view(torch.float32)
        _check(self.lib.pcs_bn_bwd_reduce_partials(_ptr(partial), partial.numel() // (2 * c), c, _ptr(buf), buf.numel(), _stream()),
               "pcs_bn_bwd_reduce_partials")
        return sums2

    def conv_gather_gemm(self, src, weight, kmap, bias=None, tile_rows=None, bn_sums=None, ordered=True, bn_raw=False, addend=None,
                         bn_bwd=None, bn_bwd_out=None, act_slope=None):
        """dst[d] = sum_{(s,d) in offset k} src[s] @ weight[k] (+bias); kmap dst-sorted. bn_sums: a list; when the
        kernel can, the [sum x | sum x^2 | n] vector of dst (what bn_stats(dst) returns) is appended to it, computed in
        the convolution's write-back instead of by a pass over dst. ordered: True = heaviest-first tile order where it
        pays (_wants_order), "force" = always, False = row order; results never depend on it."""
        src = _dev(src, "input", torch.float32)
        weight = _dev(weight, "weight", torch.float32)
        k, cin, cout = weight.shape
        if src.shape[1] != cin:
            raise ValueError("Input feature size and kernel size mismatch")  # convolution_cuda.cu:57-59
        if k != kmap.K:
            raise ValueError("kernel volume %d does not match the kernel map (%d)" % (k, kmap.K))
        if bias is not None:
            bias = _dev(bias, "bias", torch.float32)
        t = tile_rows or self.tile_rows(cin, cout, kmap)
        seg = self._segments(kmap, t)
        dst = torch.empty((kmap.n_dst, cout), dtype=torch.float32, device=src.device)
        part = self._bn_partial(kmap, t, cin, cout, k, 0, bn_sums, src.device)
        order = self._tile_order(kmap, t) if (ordered == "force" or (ordered and self._wants_order(kmap))) and kmap.n_dst > 0 and self.lib.pcs_conv_uses_tile_order(cin, cout, k, 0) else None
        # write-back extras: addend (dgrad + skip gradient), bn_bwd (the backward statistics of the BatchNorm this gradient enters)
        ep, keep, gpart = self._epilogue(addend, bn_bwd, kmap, cout, torch.float32, t, cin, k, 0, src.device, act_slope)
        _check(self.lib.pcs_conv_gather_gemm_f32_ex(_ptr(src), src.shape[0], cin, _ptr(weight), k, cout,
                                                    _ptr(kmap._pairs_raw), 0, _ptr(seg), t, kmap.n_dst,
                                                    _ptr(bias) if bias is not None else None, ep, _ptr(dst),
                                                    _ptr(gpart) if gpart is not None else (_ptr(part) if part is not None else None),
                                                    _ptr(order) if order is not None else None,
                                                    _stream()), "pcs_conv_gather_gemm_f32")
        if gpart is not None and bn_bwd_out is not None:
            bn_bwd_out.append(gpart)
        if part is not None:
            self._bn_reduce(part, t, kmap, cout, bn_sums, bn_raw)
        return dst

    # -- half-precision convolution (bf16 / fp16 MFMA) ------------------------------------------------------
    _HALF = {torch.bfloat16: 1, torch.float16: 2}

    def conv_h_applies(self, cin, cout, k):
        return bool(self.lib.pcs_conv_h_applies(int(cin), int(cout), int(k)))

    def prepare_weights_h(self, weight, dtype, transpose):
        """fp32 master weights (K, A, B) -> fragment-ordered `dtype` weights for the half kernels (uint8 buffer).
        transpose=False: forward (contract over A); True: dgrad (contract over B)."""
        weight = _dev(weight, "weight", torch.float32)
        k, a, b = weight.shape
        con, cols = (b, a) if transpose else (a, b)
        nbytes = self.lib.pcs_conv_prepared_weights_bytes(k, con, cols)
        if nbytes == 0:
            raise RuntimeError("openpcseg_amd: shape (%d, %d, %d) is not served by the half-precision kernels" % (k, a, b))
        wp = torch.empty(nbytes, dtype=torch.uint8, device=weight.device)
        _check(self.lib.pcs_conv_prepare_weights_h(_ptr(weight), k, a, b, int(bool(transpose)), self._HALF[dtype], _ptr(wp),
                                                   _stream()), "pcs_conv_prepare_weights_h")
        wp._pcs_prepared = ("half", dtype, k, con, cols)  # what the opaque buffer holds: checked by the conv call
        return wp

    def conv_gather_gemm_h(self, src, wp, k, cout, kmap, bias=None, tile_rows=None, bn_sums=None, ordered=True, bn_raw=False,
                           addend=None, bn_bwd=None, bn_bwd_out=None, act_slope=None):
        """Half-precision fused conv: src (n, cin) bf16 / fp16, wp = prepare_weights_h(...) of the same dtype."""
        if src.dtype not in self._HALF:
            raise TypeError("openpcseg_amd: conv_gather_gemm_h wants bfloat16 / float16 features, got %s" % src.dtype)
        src = _dev(src, "input")
        cin = src.shape[1]
        if k != kmap.K:
            raise ValueError("kernel volume %d does not match the kernel map (%d)" % (k, kmap.K))
        meta = getattr(wp, "_pcs_prepared", None)
        if meta is not None and meta != ("half", src.dtype, k, cin, cout):
            raise ValueError("openpcseg_amd: prepared weights %s do not match this call (%s)" % (meta, ("half", src.dtype, k, cin, cout)))
        if bias is not None:
            bias = _dev(bias, "bias", torch.float32)
        t = tile_rows or self.tile_rows(cin, cout, kmap, self._HALF[src.dtype])
        seg = self._segments(kmap, t)
        dst = torch.empty((kmap.n_dst, cout), dtype=src.dtype, device=src.device)
        part = self._bn_partial(kmap, t, cin, cout, k, self._HALF[src.dtype], bn_sums, src.device)
        order = self._tile_order(kmap, t) if (ordered == "force" or (ordered and self._wants_order(kmap))) and kmap.n_dst > 0 else None
        ep, keep, gpart = self._epilogue(addend, bn_bwd, kmap, cout, src.dtype, t, cin, k, self._HALF[src.dtype], src.device, act_slope)
        _check(self.lib.pcs_conv_gather_gemm_h_ex(_ptr(src), src.shape[0], cin, _ptr(wp), k, cout, _ptr(kmap._pairs_raw), 0,
                                                  _ptr(seg), t, kmap.n_dst, _ptr(bias) if bias is not None else None, ep,
                                                  _ptr(dst), self._HALF[src.dtype],
                                                  _ptr(gpart) if gpart is not None else (_ptr(part) if part is not None else None),
                                                  _ptr(order) if order is not None else None,
                                                  _stream()), "pcs_conv_gather_gemm_h")
        if gpart is not None and bn_bwd_out is not None:
            bn_bwd_out.append(gpart)
        if part is not None:
            self._bn_reduce(part, t, kmap, cout, bn_sums, bn_raw)
        return dst

    # -- fp32 convolution on the 16-bit MFMAs (three bf16 planes per operand, opt-in) ------------------------------
    def conv_x3_applies(self, cin, cout, k):
        return bool(self.lib.pcs_conv_x3_applies(int(cin), int(cout), int(k)))

    def prepare_weights_x3(self, weight, transpose):
        """fp32 weights (K, A, B) -> three bf16 planes in MFMA fragment order (uint8 buffer) for conv_gather_gemm_x3.
        transpose=False: forward (contract over A); True: dgrad (contract over B)."""
        weight = _dev(weight, "weight", torch.float32)
        k, a, b = weight.shape
        con, cols = (b, a) if transpose else (a, b)
        nbytes = self.lib.pcs_conv_prepared_weights_x3_bytes(k, con, cols)
        if nbytes == 0 or not self.conv_x3_applies(con, cols, k):
            raise RuntimeError("openpcseg_amd: shape (%d, %d, %d) is not served by the bf16x3 kernel" % (k, a, b))
        wp = torch.empty(nbytes, dtype=torch.uint8, device=weight.device)
        _check(self.lib.pcs_conv_prepare_weights_x3(_ptr(weight), k, a, b, int(bool(transpose)), _ptr(wp), _stream()),
               "pcs_conv_prepare_weights_x3")
        wp._pcs_prepared = ("x3", k, con, cols)
        return wp

    def conv_gather_gemm_x3(self, src, wp, k, cout, kmap, bias=None, tile_rows=None, bn_sums=None, ordered=True, bn_raw=False):
        """fp32 fused conv on the bf16 MFMAs: src (n, cin) fp32, wp = prepare_weights_x3(...); fp32 out (fp32-grade)."""
        src = _dev(src, "input", torch.float32)
        cin = src.shape[1]
        if k != kmap.K:
            raise ValueError("kernel volume %d does not match the kernel map (%d)" % (k, kmap.K))
        meta = getattr(wp, "_pcs_prepared", None)
        if meta is not None and meta != ("x3", k, cin, cout):
            raise ValueError("openpcseg_amd: prepared weights %s do not match this call (%s)" % (meta, ("x3", k, cin, cout)))
        if bias is not None:
            bias = _dev(bias, "bias", torch.float32)
        t = tile_rows or self.tile_rows(cin, cout, kmap)   # the fp32 kernels' heights (same fp32 accumulator tile, <= its width)
        seg = self._segments(kmap, t)
        dst = torch.empty((kmap.n_dst, cout), dtype=torch.float32, device=src.device)
        part = None
        if bn_sums is not None and kmap.n_dst > 0 and self.lib.pcs_conv_x3_emits_bn_partials(cin, cout, k, t):
            part = torch.empty(((kmap.n_dst + t - 1) // t) * 2 * cout, dtype=torch.float64, device=src.device)
        order = self._tile_order(kmap, t) if (ordered == "force" or (ordered and self._wants_order(kmap))) and kmap.n_dst > 0 else None
        _check(self.lib.pcs_conv_gather_gemm_f32_bf16x3(_ptr(src), src.shape[0], cin, _ptr(wp), k, cout, _ptr(kmap._pairs_raw),
                                                        0, _ptr(seg), t, kmap.n_dst, _ptr(bias) if bias is not None else None,
                                                        _ptr(dst), _ptr(part) if part is not None else None,
                                                        _ptr(order) if order is not None else None, _stream()),
               "pcs_conv_gather_gemm_f32_bf16x3")
        if part is not None:
            self._bn_reduce(part, t, kmap, cout, bn_sums, bn_raw)
        return dst

    def conv_wgrad_h(self, fa, fb, kmap, a_col):
        """Weight gradient from half operands, accumulated and returned in fp32 (K, ca, cb)."""
        if fa.dtype not in self._HALF or fb.dtype != fa.dtype:
            raise TypeError("openpcseg_amd: conv_wgrad_h wants two tensors of the same half dtype")
        fa, fb = _dev(fa, "input"), _dev(fb, "grad_output")
        ca, cb = fa.shape[1], fb.shape[1]
        gw = torch.empty((kmap.K, ca, cb), dtype=torch.float32, device=fa.device)
        ws_bytes = self.lib.pcs_conv_wgrad_ws_bytes(kmap._koff_c, kmap.K, ca, cb)
        ws = torch.empty(max(ws_bytes, 4), dtype=torch.uint8, device=fa.device)
        _check(self.lib.pcs_conv_wgrad_h(_ptr(fa), ca, _ptr(fb), cb, _ptr(kmap.pairs), a_col, _ptr(kmap.koff),
                                         kmap._koff_c, kmap.K, _ptr(gw), _ptr(ws), ws_bytes, self._HALF[fa.dtype],
                                         _stream()), "pcs_conv_wgrad_h")
        return gw

    def transpose_weights(self, w):
        """(K, A, B) -> (K, B, A) contiguous: the weights dgrad contracts with."""
        w = _dev(w, "weight", torch.float32)
        k, a, b = w.shape
        out = torch.empty((k, b, a), dtype=torch.float32, device=w.device)
        _check(self.lib.pcs_transpose_kab_f32(_ptr(w), k, a, b, _ptr(out), _stream()), "pcs_transpose_kab_f32")
        return out

    def prepared_weights_buffer(self, weight, kind, transpose):
        """An empty destination for one weights_multi job: kind "t" -> (K, B, A) fp32, a half dtype -> the fragment-ordered
        buffer of prepare_weights_h (tagged like its result)."""
        k, a, b = weight.shape
        if kind == "t":
            return torch.empty((k, b, a), dtype=torch.float32, device=weight.device)
        con, cols = (b, a) if transpose else (a, b)
        nbytes = self.lib.pcs_conv_prepared_weights_bytes(k, con, cols)
        if nbytes == 0:
            raise RuntimeError("openpcseg_amd: shape (%d, %d, %d) is not served by the half-precision kernels" % (k, a, b))
        wp = torch.empty(nbytes, dtype=torch.uint8, device=weight.device)
        wp._pcs_prepared = ("half", kind, k, con, cols)
        return wp

    def weights_multi(self, jobs):
        """jobs: [(weight (K, A, B) fp32, dst, kind, transpose)], kind "t" = transpose_weights into dst, torch.bfloat16 /
        torch.float16 = prepare_weights_h(weight, kind, transpose) into dst -- all of them in ONE launch
        (csrc/weights_multi.hip). The device copy of the job table is kept while the pointers stay the same."""
        if not jobs:
            return
        arr = (_WeightJob * len(jobs))()
        sig = []
        for j, (w, dst, kind, tr) in zip(arr, jobs):
            w = _dev(w, "weight", torch.float32)
            if not w.is_contiguous() or w.dim() != 3:
                raise ValueError("openpcseg_amd: weights_multi wants contiguous (K, A, B) weights")
            j.src, j.dst = w.data_ptr(), dst.data_ptr()
            j.K, j.A, j.B = w.shape
            j.kind = 0 if kind == "t" else self._HALF[kind]
            j.transpose = int(bool(tr))
            sig.append((j.src, j.dst, j.K, j.A, j.B, j.kind, j.transpose))
        sig = (tuple(sig), str(jobs[0][0].device))
        hit = getattr(self, "_wm_table", None)
        if hit is None or hit[0] != sig:
            blocks = self.lib.pcs_weights_multi_plan(ctypes.byref(arr), len(jobs))
            if blocks < 0:
                _check(-1, "pcs_weights_multi_plan")
            host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).pin_memory()
            hit = (sig, host.to(jobs[0][0].device, non_blocking=True), int(blocks), host)
            self._wm_table = hit
        _check(self.lib.pcs_weights_multi(_ptr(hit[1]), len(jobs), hit[2], _stream()), "pcs_weights_multi")

    def conv_wgrad(self, fa, fb, kmap, a_col, split=False):
        """gW[k] = sum_{pairs of k} fa[pair[a_col]]^T (x) fb[pair[1-a_col]] -> (K, ca, cb). split=True: the operands go
        through the bf16 MFMAs as three bf16 planes each (fp32-grade result, pcs_conv_wgrad_f32_bf16x3)."""
        fa = _dev(fa, "input", torch.float32)
        fb = _dev(fb, "grad_output", torch.float32)
        ca, cb = fa.shape[1], fb.shape[1]
        gw = torch.empty((kmap.K, ca, cb), dtype=torch.float32, device=fa.device)
        ws_bytes = self.lib.pcs_conv_wgrad_ws_bytes(kmap._koff_c, kmap.K, ca, cb)
        ws = torch.empty(max(ws_bytes, 4), dtype=torch.uint8, device=fa.device)
        fn = self.lib.pcs_conv_wgrad_f32_bf16x3 if split else self.lib.pcs_conv_wgrad_f32
        _check(fn(_ptr(fa), ca, _ptr(fb), cb, _ptr(kmap.pairs), a_col, _ptr(kmap.koff), kmap._koff_c, kmap.K, _ptr(gw),
                  _ptr(ws), ws_bytes, _stream()), "pcs_conv_wgrad_f32")
        return gw


    # -- cylinder / range scatter ---------------------------------------------------------------
    def _csr(self, index, m):
        """(order, rowptr) of a scatter index: entries sorted by target row (stable), row starts. int32 device indices
        (voxelize, devoxelize) through pcs_index_csr_i32 -- a radix sort over the bits of m only; other dtypes through torch."""
        flat = index.reshape(-1)
        if flat.is_cuda and flat.dtype == torch.int32 and os.environ.get("PCS_INDEX_CSR", "1") != "0":
            flat = flat if flat.is_contiguous() else flat.contiguous()
            n = flat.numel()
            order = torch.empty(n, dtype=torch.int64, device=flat.device)
            rowptr = torch.empty(m + 1, dtype=torch.int64, device=flat.device)
            ws_bytes = self.lib.pcs_index_csr_ws_bytes(n, m)
            ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=flat.device)
            _check(self.lib.pcs_index_csr_i32(_ptr(flat), n, m, _ptr(order), _ptr(rowptr), _ptr(ws), ws_bytes, _stream()),
                   "pcs_index_csr_i32")
            return order, rowptr
        vals, order = torch.sort(flat)
        rowptr = torch.searchsorted(vals, torch.arange(m + 1, device=index.device, dtype=vals.dtype))
        return order.contiguous(), rowptr.contiguous()

    def scatter_max_fwd(self, src, index, m):
        src = _dev(src, "src", torch.float32)
        index = _dev(index, "index", torch.int64)
        c = src.shape[1]
        # the cylinder models scatter several tensors over one index
        order, rowptr = _cached(index, "_pcs_csr", _cache_key(index) + (m,), lambda: self._csr(index, m))
        out = torch.empty((m, c), dtype=torch.float32, device=src.device)
        arg = torch.empty((m, c), dtype=torch.int32, device=src.device)
        _check(self.lib.pcs_scatter_max_fwd_f32(_ptr(src), _ptr(order), _ptr(rowptr), m, c, _ptr(out), _ptr(arg),
                                                _stream()), "pcs_scatter_max_fwd_f32")
        return out, arg

    def scatter_max_bwd(self, gout, arg, n):
        gout = _dev(gout, "grad_output", torch.float32)
        m, c = gout.shape
        gsrc = torch.empty((n, c), dtype=torch.float32, device=gout.device)
        _check(self.lib.pcs_scatter_max_bwd_f32(_ptr(gout), _ptr(arg), m, n, c, _ptr(gsrc), _stream()),
               "pcs_scatter_max_bwd_f32")
        return gsrc

    def map_count(self, pxpy, b, h, w):
        pxpy = _dev(pxpy, "pxpy", torch.int32)
        out = torch.empty((b, h, w), dtype=torch.int32, device=pxpy.device)
        _check(self.lib.pcs_map_count(_ptr(pxpy), pxpy.shape[0], b, h, w, _ptr(out), _stream()), "pcs_map_count")
        return out

    def _pixel_csr(self, pxpy, b, h, w):
        """Points sorted by pixel (out-of-image points first) + row pointers over the B*H*W pixels; cached on pxpy."""
        def make():
            pb, px, py = pxpy[:, 0].long(), pxpy[:, 1].long(), pxpy[:, 2].long()
            ok = (pb >= 0) & (pb < b) & (px >= 0) & (px < w) & (py >= 0) & (py < h)
            key = torch.where(ok, (pb * h + py) * w + px, torch.full_like(pb, -1))
            return self._csr(key, b * h * w)
        return _cached(pxpy, "_pcs_px_csr", _cache_key(pxpy) + (b, h, w), make)

    def denselize_fwd(self, feat, count_map, pxpy):
        """out[b, :, py, px] = mean of the feature rows of the points of pixel (b, py, px). C % 4 == 0: segmented over
        the points sorted by pixel, NCHW stores in full lines, no atomics; otherwise the reference's atomic dataflow."""
        feat = _dev(feat, "feat", torch.float32)
        count_map = _dev(count_map, "count_map", torch.int32)
        pxpy = _dev(pxpy, "pxpy", torch.int32)
        (b, h, w), c = count_map.shape, feat.shape[1]
        if c % 4:
            return self.denselize_fwd_atomic(feat, count_map, pxpy)
        csr = self._pixel_csr(pxpy, b, h, w)
        out = torch.empty((b, c, h, w), dtype=torch.float32, device=feat.device)
        _check(self.lib.pcs_denselize_fwd_csr_f32(_ptr(feat), _ptr(csr[0]), _ptr(csr[1]), _ptr(count_map), b, c, h, w,
                                                  _ptr(out), _stream()), "pcs_denselize_fwd_csr_f32")
        return out

    def denselize_fwd_atomic(self, feat, count_map, pxpy):
        """The reference's literal K14 dataflow (NCHW fp32 atomics): odd channel counts and A/B measurements."""
        feat = _dev(feat, "feat", torch.float32)
        count_map = _dev(count_map, "count_map", torch.int32)
        pxpy = _dev(pxpy, "pxpy", torch.int32)
        (b, h, w), c = count_map.shape, feat.shape[1]
        out = torch.empty((b, c, h, w), dtype=torch.float32, device=feat.device)
        _check(self.lib.pcs_denselize_fwd_f32(_ptr(feat), _ptr(count_map), _ptr(pxpy), feat.shape[0], b, c, h, w,
                                              _ptr(out), _stream()), "pcs_denselize_fwd_f32")
        return out

    def denselize_bwd(self, gout, count_map, pxpy):
        gout = _dev(gout, "top_grad", torch.float32)
        b, c, h, w = gout.shape
        n = pxpy.shape[0]
        gfeat = torch.empty((n, c), dtype=torch.float32, device=gout.device)
        if c % 4:
            return self.denselize_bwd_gather(gout, count_map, pxpy)
        csr = self._pixel_csr(pxpy, b, h, w)
        _check(self.lib.pcs_denselize_bwd_csr_f32(_ptr(gout), _ptr(csr[0]), _ptr(csr[1]), _ptr(count_map), n, b, c, h, w,
                                                  _ptr(gfeat), _stream()), "pcs_denselize_bwd_csr_f32")
        return gfeat

    def denselize_bwd_gather(self, gout, count_map, pxpy):
        """The reference's K15 dataflow (per-point strided NCHW gather): odd channel counts and A/B measurements."""
        gout = _dev(gout, "top_grad", torch.float32)
        b, c, h, w = gout.shape
        n = pxpy.shape[0]
        gfeat = torch.empty((n, c), dtype=torch.float32, device=gout.device)
        _check(self.lib.pcs_denselize_bwd_f32(_ptr(gout), _ptr(count_map), _ptr(pxpy), n, b, c, h, w, _ptr(gfeat),
                                              _stream()), "pcs_denselize_bwd_f32")
        return gfeat

    # -- range image -> points (RPVNet's range_to_point) ------------------------------------------------------
    def range_sample_fwd(self, img, pxpy):
        """(n, C) bilinear samples of img (B, C, H, W) at pxpy (n, 3) = (frame, x, y) -- grid_sample(bilinear, zeros, align_corners=False)
        for every frame in one launch (R:pcseg/model/segmentor/fusion/rpvnet/rpvnet.py:31-51)."""
        img = _dev(img, "feature_map", torch.float32)
        pxpy = _dev(pxpy, "pxpy", torch.float32)
        b, c, h, w = img.shape
        n = pxpy.shape[0]
        out = torch.empty((n, c), dtype=torch.float32, device=img.device)
        _check(self.lib.pcs_range_sample_fwd_f32(_ptr(img), _ptr(pxpy), n, b, c, h, w, _ptr(out), _stream()), "pcs_range_sample_fwd_f32")
        return out

    def _corner_csr(self, pxpy, b, h, w):
        """(order, rowptr, wts) of the 4 n (point, corner) entries over the B H W pixels; cached on pxpy per resolution."""
        def make():
            n = pxpy.shape[0]
            keys = torch.empty(4 * n, dtype=torch.int64, device=pxpy.device)
            wts = torch.empty(4 * n, dtype=torch.float32, device=pxpy.device)
            _check(self.lib.pcs_range_sample_corners(_ptr(pxpy), n, b, h, w, _ptr(keys), _ptr(wts), _stream()), "pcs_range_sample_corners")
            vals, order = torch.sort(keys, stable=True)   # stable: a pixel's entries stay in (point, corner) order -> fixed summation order
            rowptr = torch.searchsorted(vals, torch.arange(b * h * w + 1, device=pxpy.device, dtype=torch.int64))
            return order.contiguous(), rowptr.contiguous(), wts
        return _cached(pxpy, "_pcs_corner_csr", _cache_key(pxpy) + (b, h, w), make)

    def range_sample_bwd(self, gout, pxpy, b, h, w):
        """d img (B, C, H, W) of range_sample_fwd: atomic-free, every element written once."""
        gout = _dev(gout, "grad_output", torch.float32)
        pxpy = _dev(pxpy, "pxpy", torch.float32)
        c = gout.shape[1]
        order, rowptr, wts = self._corner_csr(pxpy, b, h, w)
        gimg = torch.empty((b, c, h, w), dtype=torch.float32, device=gout.device)
        _check(self.lib.pcs_range_sample_bwd_csr_f32(_ptr(gout), _ptr(order), _ptr(rowptr), _ptr(wts), b, c, h, w, _ptr(gimg),
                                                     _stream()), "pcs_range_sample_bwd_csr_f32")
        return gimg

    # -- fused BatchNorm (+residual, +ReLU) -------------------------------------------------------
    @staticmethod
    def _feat(t, name, like=None):
        """A feature tensor of the BN passes: fp32, bf16 or fp16 on the device (and of `like`'s dtype when given)."""
        t = _dev(t, name)
        if t.dtype not in (torch.float32, torch.bfloat16, torch.float16):
            raise TypeError("openpcseg_amd: `%s` must be float32 / bfloat16 / float16, got %s" % (name, t.dtype))
        if like is not None and t.dtype != like.dtype:
            t = t.to(like.dtype)
        return t

    def bn_stats(self, x):
        """-> sums (2c + 1,) float64 = [sum x | sum x^2 | n] (the vector SyncBN all-reduces; the count rides along)."""
        x = self._feat(x, "input")
        n, c = x.shape
        ws = torch.empty(self.lib.pcs_bn_num_partials() * 2 * c, dtype=torch.float32, device=x.device)
        sums = torch.empty(2 * c + 1, dtype=torch.float64, device=x.device)
        if x.dtype == torch.float32:
            _check(self.lib.pcs_bn_stats_f32(_ptr(x), n, c, _ptr(ws), _ptr(sums), _stream()), "pcs_bn_stats_f32")
        else:
            _check(self.lib.pcs_bn_stats_h(_ptr(x), n, c, self._HALF[x.dtype], _ptr(ws), _ptr(sums), _stream()),
                   "pcs_bn_stats_h")
        return sums

    def bn_finalize(self, sums, count, eps, momentum, running_mean, running_var, count_dev=None):
        """stat = [mean | invstd]; `count_dev` (1-element float64 device tensor, e.g. sums[2c:]) replaces the host count."""
        c = sums.numel() // 2
        stat = torch.empty(2 * c, dtype=torch.float64, device=sums.device)
        _check(self.lib.pcs_bn_finalize_f32(_ptr(sums), float(count), _ptr(count_dev) if count_dev is not None else None,
                                            c, float(eps), float(momentum),
                                            _ptr(running_mean) if running_mean is not None else None,
                                            _ptr(running_var) if running_var is not None else None, _ptr(stat),
                                            _stream()), "pcs_bn_finalize_f32")
        return stat

    def bn_apply(self, x, res, stat, w, b, relu, want_mask=False, tail=None):
        """y = act((x - mean) * invstd * w + b [+ res]) in x's dtype (fp32 / bf16 / fp16). want_mask (c % 32 == 0): also
        the ReLU gate as n x c/32 int32 words (bit ch % 32 of word ch / 32 = [y > 0]) -- what the backward passes read
        instead of y. tail (n, ct): concat fusion -- the result is the (n, c + ct) tensor cat([y, tail], 1), y written
        straight into its left columns and `tail` copied to the right ones by the same launch."""
        x = self._feat(x, "input")
        res = self._feat(res, "residual", x) if res is not None else None
        tail = self._feat(tail, "tail", x) if tail is not None else None
        n, c = x.shape
        ct = tail.shape[1] if tail is not None else 0
        if want_mask and c % 32:
            raise RuntimeError("openpcseg_amd: the ReLU bit mask needs a channel count that is a multiple of 32")
        if tail is not None and (tail.shape[0] != n or c % 4 or ct % 4):
            raise RuntimeError("openpcseg_amd: concat fusion needs equal row counts and channel counts that are multiples of 4")
        y = torch.empty((n, c + ct), dtype=x.dtype, device=x.device)
        mask = torch.empty((n, c // 32), dtype=torch.int32, device=x.device) if want_mask else None
        args = [_ptr(x), _ptr(res) if res is not None else None, _ptr(stat), _ptr(w) if w is not None else None,
                _ptr(b) if b is not None else None, n, c, int(relu)]
        cat = [c + ct, _ptr(tail) if tail is not None else None, ct]
        if x.dtype == torch.float32:
            _check(self.lib.pcs_bn_apply_f32(*args, _ptr(y), _ptr(mask) if want_mask else None, *cat, _stream()),
                   "pcs_bn_apply_f32")
        else:
            _check(self.lib.pcs_bn_apply_h(*args, self._HALF[x.dtype], _ptr(y), _ptr(mask) if want_mask else None, *cat,
                                           _stream()), "pcs_bn_apply_h")
        return (y, mask) if want_mask else y

    @staticmethod
    def _gate(gate, relu):
        """(y pointer, mask pointer) of the ReLU gate: the output tensor (features dtype) or its bit mask (int32)."""
        if not relu or gate is None:
            return None, None
        return (None, _ptr(gate)) if gate.dtype == torch.int32 else (_ptr(gate), None)

    @staticmethod
    def _rows(t, name, like, c):
        """dy of the BN backward passes: (n, c) in like's dtype, unit column stride, any row stride that keeps the rows
        vector-aligned (a column slice of the gradient of a concat buffer); anything else is made contiguous."""
        if not isinstance(t, torch.Tensor) or not t.is_cuda:
            raise RuntimeError("openpcseg_amd: `%s` must be a HIP device tensor" % name)
        if t.dtype != like.dtype:
            t = t.to(like.dtype)
        unit = 16 // t.element_size() if t.dtype == torch.float32 else 4
        ok = t.dim() == 2 and t.stride(1) == 1 and t.stride(0) >= c and t.stride(0) % 4 == 0 and \
            t.data_ptr() % (unit * t.element_size()) == 0
        if not ok:
            t = t.contiguous()
        return t, t.stride(0)

    def bn_bwd_stats(self, dy, x, gate, stat, relu):
        x = self._feat(x, "input")
        n, c = x.shape
        dy, lddy = self._rows(dy, "grad_output", x, c)
        ws = torch.empty(self.lib.pcs_bn_num_partials() * 2 * c, dtype=torch.float32, device=x.device)
        buf = torch.empty(3 * c, dtype=torch.float64, device=x.device)   # 2c doubles, then the same 2c values as floats
        sums2 = buf[:2 * c]
        sums2._pcs_f32 = buf[2 * c:].view(torch.float32)   # [sum g | sum g xhat] in fp32: db | dw without a conversion launch
        yp, mp = self._gate(gate, relu)
        if x.dtype == torch.float32:
            _check(self.lib.pcs_bn_bwd_stats_f32(_ptr(dy), _ptr(x), yp, mp, _ptr(stat), n, c, int(relu),
                                                 _ptr(ws), _ptr(sums2), buf.numel(), lddy, _stream()), "pcs_bn_bwd_stats_f32")
        else:
            _check(self.lib.pcs_bn_bwd_stats_h(_ptr(dy), _ptr(x), yp, mp, _ptr(stat), n, c, int(relu), self._HALF[x.dtype],
                                               _ptr(ws), _ptr(sums2), buf.numel(), lddy, _stream()), "pcs_bn_bwd_stats_h")
        return sums2

    def bn_bwd_apply(self, dy, x, gate, stat, sums2, count, w, relu, want_res, count_dev=None, in_slope=None):
        x = self._feat(x, "input")
        n, c = x.shape
        dy, lddy = self._rows(dy, "grad_output", x, c)
        dx = torch.empty_like(x)
        dres = torch.empty_like(x) if want_res else None
        yp, mp = self._gate(gate, relu)
        if in_slope is not None and float(in_slope) != 1.0:   # x is a LeakyReLU output: dx leaves as the gradient of the pre-activation
            code = 0 if x.dtype == torch.float32 else self._HALF[x.dtype]
            _check(self.lib.pcs_bn_bwd_apply_act(_ptr(dy), _ptr(x), yp, mp, _ptr(stat), _ptr(sums2), float(count),
                                                 _ptr(count_dev) if count_dev is not None else None,
                                                 _ptr(w) if w is not None else None, n, c, int(relu), code, float(in_slope),
                                                 _ptr(dx), _ptr(dres) if want_res else None, lddy, _stream()), "pcs_bn_bwd_apply_act")
            return dx, dres
        head = [_ptr(dy), _ptr(x), yp, mp, _ptr(stat), _ptr(sums2), float(count),
                _ptr(count_dev) if count_dev is not None else None, _ptr(w) if w is not None else None, n, c, int(relu)]
        tail = [_ptr(dx), _ptr(dres) if want_res else None, lddy, _stream()]
        if x.dtype == torch.float32:
            _check(self.lib.pcs_bn_bwd_apply_f32(*head, *tail), "pcs_bn_bwd_apply_f32")
        else:
            _check(self.lib.pcs_bn_bwd_apply_h(*head, self._HALF[x.dtype], *tail), "pcs_bn_bwd_apply_h")
        return dx, dres

    # -- device-side sparse_quantize --------------------------------------------------------------------
    def quantize(self, points, voxel_size3, want_index, want_inverse):
        """points (n, >=3) float32 / int32 on the device -> (vox (m,3) int32, index (m) int64 | None,
        inverse (n) int64 | None); reference order and representative (TS:torchsparse/utils/quantize.py:24-46)."""
        points = _dev(points, "coords")
        if points.dim() != 2 or points.shape[1] != 3:
            # the reference divides (n, d) by a 3-vector (quantize.py:33): NumPy broadcasting admits d == 3 only
            raise ValueError("sparse_quantize: coords must be (n, 3), got %s" % (tuple(points.shape),))
        if points.dtype not in (torch.float32, torch.float64, torch.int32):
            # half / bfloat16 -> float32 (exact); other integer types -> int32. float64 stays float64: NumPy divides
            # float64 by float64, and a float32 round trip moves points that sit near a voxel face
            points = points.float() if points.is_floating_point() else points.int()
        points = points.contiguous()
        kind = 2 if points.dtype == torch.float64 else int(points.is_floating_point())
        n, stride = points.shape
        dev = points.device
        vox_size = (c_double * 3)(*[float(v) for v in voxel_size3])
        coords = torch.empty((n, 3), dtype=torch.int32, device=dev)
        i32 = torch.iinfo(torch.int32)
        bbox = _h2d([i32.max] * 3 + [i32.min] * 3, torch.int32, dev)
        _check(self.lib.pcs_quantize_floor(_ptr(points), kind, n, stride, vox_size,
                                           _ptr(coords), _ptr(bbox), _stream()), "pcs_quantize_floor")
        keys = torch.empty(n, dtype=torch.int64, device=dev)
        _check(self.lib.pcs_quantize_keys(_ptr(coords), n, _ptr(bbox), _ptr(keys), _stream()), "pcs_quantize_keys")
        skeys, perm = torch.sort(keys, stable=True)  # rocPRIM radix sort: equal keys keep row order
        flags = torch.empty(n, dtype=torch.int32, device=dev)
        _check(self.lib.pcs_quantize_flags(_ptr(skeys), n, _ptr(flags), _stream()), "pcs_quantize_flags")
        rank = torch.cumsum(flags, dim=0, dtype=torch.int64)
        m = int(rank[-1].item()) if n else 0  # the one host sync: the number of voxels sizes the outputs
        vox = torch.empty((m, 3), dtype=torch.int32, device=dev)
        index = torch.empty(m, dtype=torch.int64, device=dev) if want_index else None
        inverse = torch.empty(n, dtype=torch.int64, device=dev) if want_inverse else None
        _check(self.lib.pcs_quantize_emit(_ptr(flags), _ptr(rank), _ptr(perm), _ptr(coords), n, _ptr(vox),
                                          _ptr(index) if want_index else None,
                                          _ptr(inverse) if want_inverse else None, _stream()), "pcs_quantize_emit")
        return vox, index, inverse


    def quantize_frame_keys(self, coords, frames):
        """int64 sort keys of a collated batch: frame on top of the batch's bounding box (one launch + two reductions)."""
        coords = _dev(coords, "coords", torch.int32)
        frames = _dev(frames, "frames", torch.int64)
        n = coords.shape[0]
        keys = torch.empty(n, dtype=torch.int64, device=coords.device)
        if n == 0:
            return keys
        bbox = torch.cat([coords.amin(0), coords.amax(0)]).contiguous()
        _check(self.lib.pcs_quantize_frame_keys(_ptr(coords), _ptr(frames), n, _ptr(bbox), _ptr(keys), _stream()), "pcs_quantize_frame_keys")
        return keys

    def quantize_sorted_keys(self, keys, coords, frames):
        """Voxel dedup of a whole batch from ready-made keys (hostdata.sparse_quantize_frames): one stable sort, flags, scan,
        emit -> (vox (m, 4) int32 [x, y, z, frame], index (m,), inverse (n,))."""
        keys = _dev(keys, "keys", torch.int64)
        coords = _dev(coords, "coords", torch.int32)
        n = keys.numel()
        dev = keys.device
        if n == 0:
            return (torch.empty((0, 4), dtype=torch.int32, device=dev), torch.empty(0, dtype=torch.int64, device=dev),
                    torch.empty(0, dtype=torch.int64, device=dev))
        skeys, perm = torch.sort(keys, stable=True)
        flags = torch.empty(n, dtype=torch.int32, device=dev)
        _check(self.lib.pcs_quantize_flags(_ptr(skeys), n, _ptr(flags), _stream()), "pcs_quantize_flags")
        rank = torch.cumsum(flags, dim=0, dtype=torch.int64)
        m = int(rank[-1].item())  # the one host read of the batch
        vox = torch.empty((m, 3), dtype=torch.int32, device=dev)
        index = torch.empty(m, dtype=torch.int64, device=dev)
        inverse = torch.empty(n, dtype=torch.int64, device=dev)
        _check(self.lib.pcs_quantize_emit(_ptr(flags), _ptr(rank), _ptr(perm), _ptr(coords), n, _ptr(vox), _ptr(index),
                                          _ptr(inverse), _stream()), "pcs_quantize_emit")
        return torch.cat([vox, frames[index].int()[:, None]], dim=1), index, inverse

    def unique_inverse_csr(self, keys):
        """keys (n,) int64 -> (uniq (m,) ascending, inverse (n,) int64, counts (m,) int32); the stable sort behind it is
        kept as the segmented-reduction CSR of `inverse` (cached on the returned tensor, so spvoxelize over it sorts
        nothing). One radix sort + three streaming passes for torch.unique + sphashquery + spcount."""
        keys = _dev(keys, "keys", torch.int64)
        n = keys.numel()
        dev = keys.device
        if n == 0:
            z = torch.empty(0, dtype=torch.int64, device=dev)
            return z, z.clone(), torch.empty(0, dtype=torch.int32, device=dev)
        skeys, perm = torch.sort(keys, stable=True)
        flags = torch.empty(n, dtype=torch.int32, device=dev)
        _check(self.lib.pcs_quantize_flags(_ptr(skeys), n, _ptr(flags), _stream()), "pcs_quantize_flags")
        rank = torch.cumsum(flags, dim=0, dtype=torch.int64)
        m = int(rank[-1].item())  # the one host sync (torch.unique has the same one): sizes the outputs
        uniq = torch.empty(m, dtype=torch.int64, device=dev)
        inverse = torch.empty(n, dtype=torch.int64, device=dev)
        rowptr = torch.empty(m + 1, dtype=torch.int64, device=dev)
        _check(self.lib.pcs_unique_emit(_ptr(flags), _ptr(rank), _ptr(perm), _ptr(skeys), n, _ptr(uniq), _ptr(inverse),
                                        _ptr(rowptr), _stream()), "pcs_unique_emit")
        counts = (rowptr[1:] - rowptr[:-1]).int()
        inverse._pcs_vox_csr = (_cache_key(inverse) + (m, n), (perm, rowptr))  # what voxelize_fwd would sort for
        return uniq, inverse, counts

    # -- cylinder front-end ------------------------------------------------------------------------------
    def cylinder_partition(self, points, space_min, space_max, grid_size, want_polar=True, want_feat=True):
        """points (n, >=3) float32 -> (polar (n,3) f32 | None, coord (n,3) int32, feat (n, 8 + extras) f32 | None)."""
        points = _dev(points, "points", torch.float32)
        n, stride = points.shape
        lo = (c_double * 3)(*[float(v) for v in space_min])
        hi = (c_double * 3)(*[float(v) for v in space_max])
        grid = (c_int32 * 3)(*[int(v) for v in grid_size])
        dev = points.device
        polar = torch.empty((n, 3), dtype=torch.float32, device=dev) if want_polar else None
        coord = torch.empty((n, 3), dtype=torch.int32, device=dev)
        feat = torch.empty((n, 8 + stride - 3), dtype=torch.float32, device=dev) if want_feat else None
        _check(self.lib.pcs_cylinder_partition_f32(_ptr(points), n, stride, lo, hi, grid,
                                                   _ptr(polar) if want_polar else None, _ptr(coord),
                                                   _ptr(feat) if want_feat else None, _stream()),
               "pcs_cylinder_partition_f32")
        return polar, coord, feat

    def voxel_label_vote(self, inverse, labels, m, num_classes, ignore_label):
        """-> (voxel_labels (m,) int64, bad (1,) int32 device flag: a counted label was out of range)."""
        inverse = _dev(inverse, "inverse_map", torch.int64)
        labels = _dev(labels, "point_labels", torch.int64)
        dev = inverse.device
        ws = torch.empty(max(int(m) * int(num_classes), 1), dtype=torch.int32, device=dev)
        bad = torch.empty(1, dtype=torch.int32, device=dev)
        out = torch.empty(int(m), dtype=torch.int64, device=dev)
        _check(self.lib.pcs_voxel_label_vote(_ptr(inverse), _ptr(labels), inverse.numel(), int(m), int(num_classes),
                                             int(ignore_label), _ptr(ws), _ptr(bad), _ptr(out), _stream()),
               "pcs_voxel_label_vote")
        return out, bad

    def rows_argmax_gather(self, logits, inverse=None):
        logits = _dev(logits, "logits", torch.float32)
        m, c = logits.shape
        if inverse is not None:
            inverse = _dev(inverse, "inverse_map", torch.int64)
        n = inverse.numel() if inverse is not None else m
        out = torch.empty(n, dtype=torch.int64, device=logits.device)
        _check(self.lib.pcs_rows_argmax_gather_f32(_ptr(logits), m, c, _ptr(inverse) if inverse is not None else None,
                                                   n, _ptr(out), _stream()), "pcs_rows_argmax_gather_f32")
        return out

    # -- criterion tail ---------------------------------------------------------------------------
    def lovasz_softmax(self, probas, labels, ignore=None, need_grad=True):
        """-> (loss 0-dim float32, d loss / d probas (n, C) float32 or None); csrc/lovasz.hip."""
        probas = _dev(probas, "probas", torch.float32)
        labels = _dev(labels, "labels", torch.int64)
        n, nc = probas.shape
        if labels.numel() != n:
            raise ValueError("openpcseg_amd: %d labels for %d probability rows" % (labels.numel(), n))
        has_ignore, ign = (0, 0) if ignore is None else (1, int(ignore))
        ws_bytes = self.lib.pcs_lovasz_workspace_bytes(n, nc, has_ignore, ign)
        if ws_bytes < 0:
            _check(-1, "pcs_lovasz_workspace_bytes")
        ws = torch.empty(max(int(ws_bytes), 1), dtype=torch.uint8, device=probas.device)
        loss = torch.empty((), dtype=torch.float32, device=probas.device)
        grad = torch.empty_like(probas) if need_grad else None
        _check(self.lib.pcs_lovasz_softmax_f32(_ptr(probas), _ptr(labels), n, nc, has_ignore, ign, _ptr(loss),
                                               _ptr(grad) if grad is not None else None, _ptr(ws), int(ws_bytes), _stream()),
               "pcs_lovasz_softmax_f32")
        return loss, grad


_BACKEND = None


def backend():
    """The process-wide native backend (HIP only)."""
    global _BACKEND
    if _BACKEND is None:
        _BACKEND = HipBackend()
    return _BACKEND
