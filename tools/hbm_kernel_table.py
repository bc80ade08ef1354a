#!/usr/bin/env python3
"""Achieved HBM GB/s of the bandwidth-bound ops of the hot path (hash, table, rulebook, voxelize, devoxelize,
scatter_max, denselize, fused BatchNorm) at the bench size: 12 synthetic scans = 1.44 M points / 1.16 M voxels.
bytes = ALGORITHMIC bytes of SURVEY.md section 8d (each tensor read / written once), time = HIP events around
`reps` back-to-back calls of the C-ABI entry through the Python host layer.
Usage: python tools/hbm_kernel_table.py [frames] [reps]   (markdown table on stdout)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from openpcseg_amd import functional as F  # noqa: E402
from openpcseg_amd import native  # noqa: E402
from openpcseg_amd.sparse import get_kernel_offsets  # noqa: E402
from openpcseg_amd.workloads.synthetic import make_batch  # noqa: E402

PEAK = 8000.0  # GB/s, MI355X HBM3E (MI355X_MICROARCH.md)


DRY = os.environ.get("PCS_TABLE_DRYRUN") == "1"  # CPU dry run of this script on the pure-PyTorch CPU backend (checks the calls only)


def timed(fn, reps):
    if DRY:
        fn()
        return 1.0
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps  # us


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    dev = torch.device("cpu" if DRY else "cuda:0")
    if DRY:
        from openpcseg_amd.cpu_fallback import TorchCpuBackend   # the package's explicit pure-PyTorch CPU backend
        native._BACKEND = TorchCpuBackend()
    be = native.backend()
    batch = make_batch(list(range(frames)))
    vox = batch["lidar"].C.to(dev).int().contiguous()      # (M,4) voxel coords, one row per voxel
    m = vox.shape[0]
    g = torch.Generator(device="cpu").manual_seed(0)
    n = int(m * 1.24)                                       # points: 120k rays -> ~96.6k voxels per scan
    src = torch.randint(0, m, (n,), generator=g)
    pts_vox = vox[src.to(dev)]                               # voxel of every point
    pts_f = pts_vox[:, :3].float() + torch.rand(n, 3, generator=g).to(dev)
    rows = []

    def add(name, entry, us, nbytes, note=""):
        gbs = nbytes / us / 1e3
        rows.append((name, entry, us, nbytes / 1e6, gbs, gbs / PEAK, note))

    # ---- hashing / table / rulebook -------------------------------------------------------------------
    offs = get_kernel_offsets(3, 1, 1, device=dev)
    add("sphash", "pcs_hash", timed(lambda: be.hash(vox), reps), 24 * m)
    add("sphash + offsets (K=27)", "pcs_kernel_hash", timed(lambda: be.kernel_hash(vox, offs), reps), 16 * m + 8 * 27 * m)
    keys = be.hash(vox)
    q = be.hash(pts_vox)
    if not DRY:  # the CPU backend has no persistent table object
        add("hash table build", "pcs_hashtable_build", timed(lambda: be.table_build(keys), reps), 16 * m,
            "12 B/slot table, load <= 0.5")
        table = be.table_build(keys)
        add("hash table query", "pcs_hashtable_query", timed(lambda: be.table_query(table, q), reps), 16 * n,
            "random probes into a %d MB table" % (table.capacity * 12 // 2**20))
    km = be.build_kmap(vox, vox, offs)
    p = km.num_pairs
    add("rulebook k3 s1 (probe+scan+fill)", "pcs_rulebook_*", timed(lambda: be.build_kmap(vox, vox, offs), reps),
        16 * m + 4 * 27 * m + 8 * p + 16 * m, "includes table build and the one host sync for the sizes")
    add("spdownsample s2", "pcs_downsample_pack/unpack", timed(lambda: F.spdownsample(vox, 2, 2, 1), reps), 24 * m,
        "dominated by the radix sort + unique of the packed keys (rocPRIM)")
    raw = (pts_vox[:, :3].contiguous())
    add("sparse_quantize (device, index+inverse)", "pcs_quantize_*", timed(lambda: be.quantize(raw, (1, 1, 1), True, True), reps),
        12 * n + 8 * n + 20 * m, "4 streaming passes + stable radix sort of (key,row) + scan; one host sync (voxel count)")
    idx = be.hash_query(q, keys).int()
    add("spcount", "pcs_count", timed(lambda: be.count(idx, m), reps), 4 * n + 4 * m)

    # ---- point <-> voxel ---------------------------------------------------------------------------------
    counts = be.count(idx, m)
    for c in (4, 32, 96):
        f = torch.randn(n, c, device=dev)
        add("spvoxelize C=%d" % c, "pcs_voxelize_fwd_csr_f32", timed(lambda: be.voxelize_fwd(f, idx, counts), reps),
            4 * c * (n + m) + 4 * n + 4 * m, "sorted point order cached on the index tensor")
        if not DRY:
            add("spvoxelize C=%d, atomic form" % c, "pcs_voxelize_fwd_f32", timed(lambda: be.voxelize_fwd_atomic(f, idx, counts), reps),
                4 * c * (n + m) + 4 * n + 4 * m, "reference dataflow, kept for A/B")
        gv = torch.randn(m, c, device=dev)
        add("spvoxelize bwd C=%d" % c, "pcs_voxelize_bwd_f32", timed(lambda: be.voxelize_bwd(gv, idx, counts, n), reps),
            4 * c * (n + m) + 4 * n + 4 * m)
    offs2 = get_kernel_offsets(2, 1, 1, device=dev)
    pc = torch.cat([torch.floor(pts_f).int(), pts_vox[:, 3:4]], 1).contiguous()
    idx8 = be.hash_query(be.kernel_hash(pc, offs2).view(-1), keys).view(8, n)
    w8 = be.ti_weights(pts_f, idx8, 1)
    add("calc_ti_weights", "pcs_ti_weights_f32", timed(lambda: be.ti_weights(pts_f, idx8, 1), reps), 76 * n)
    idx8t, w8t = idx8.t().contiguous().int(), w8.t().contiguous()
    for c in (32, 96, 256):
        fv = torch.randn(m, c, device=dev)
        add("spdevoxelize C=%d" % c, "pcs_devoxelize_fwd_f32", timed(lambda: be.devoxelize_fwd(fv, idx8t, w8t), reps),
            4 * c * (m + n) + 64 * n, "compulsory bytes; 8 gathered rows / point come from L2")
        gp = torch.randn(n, c, device=dev)
        be.devoxelize_bwd(gp, idx8t, w8t, m)  # builds + caches the CSR
        add("spdevoxelize bwd C=%d" % c, "pcs_devoxelize_bwd_csr_f32", timed(lambda: be.devoxelize_bwd(gp, idx8t, w8t, m), reps),
            4 * c * (m + n) + 64 * n, "CSR cached on the index tensor")

    # ---- cylinder / range ----------------------------------------------------------------------------------
    c = 256
    f = torch.randn(n, c, device=dev)
    idx64 = idx.long()
    be.scatter_max_fwd(f, idx64, m)
    add("scatter_max C=256", "pcs_scatter_max_fwd_f32", timed(lambda: be.scatter_max_fwd(f, idx64, m), reps),
        4 * c * (n + m) + 8 * n, "sorted order cached on the index tensor")
    b, h, w_ = frames, 64, 2048
    pxpy = torch.stack([torch.randint(0, b, (n,), generator=g), torch.randint(0, w_, (n,), generator=g),
                        torch.randint(0, h, (n,), generator=g)], 1).int().to(dev)
    cm = be.map_count(pxpy, b, h, w_)
    add("range map_count", "pcs_map_count", timed(lambda: be.map_count(pxpy, b, h, w_), reps), 12 * n + 4 * b * h * w_)
    f32 = torch.randn(n, 32, device=dev)
    add("range denselize C=32", "pcs_denselize_fwd_csr_f32", timed(lambda: be.denselize_fwd(f32, cm, pxpy), reps),
        4 * 32 * (n + b * h * w_) + 12 * n, "sorted point order cached on the pxpy tensor")
    if not DRY:
        add("range denselize C=32, atomic form", "pcs_denselize_fwd_f32", timed(lambda: be.denselize_fwd_atomic(f32, cm, pxpy), reps),
            4 * 32 * (n + b * h * w_) + 12 * n, "NCHW atomics (reference dataflow), kept for A/B")
    g32 = torch.randn(b, 32, h, w_, device=dev)
    add("range denselize bwd C=32", "pcs_denselize_bwd_csr_f32", timed(lambda: be.denselize_bwd(g32, cm, pxpy), reps),
        4 * 32 * (n + b * h * w_) + 12 * n)
    if not DRY:
        add("range denselize bwd C=32, gather form", "pcs_denselize_bwd_f32", timed(lambda: be.denselize_bwd_gather(g32, cm, pxpy), reps),
            4 * 32 * (n + b * h * w_) + 12 * n, "reference dataflow, kept for A/B")

    # ---- [r5] range image -> points (csrc/rangesample.hip) at RPVNet's shapes (4 frames of ~97 k points) ---------------------------
    if not DRY:
        nb = 4
        npnt = min(n, 390000)
        fpx = torch.cat([torch.sort(torch.randint(0, nb, (npnt,), generator=g)).values.float()[:, None],
                         torch.rand(npnt, 2, generator=g) * 2 - 1], 1).to(dev).contiguous()
        for c, hh, ww in ((56, 64, 2048), (168, 64, 2048), (224, 16, 512), (448, 4, 128)):
            img = torch.randn(nb, c, hh, ww, device=dev)
            gp = torch.randn(npnt, c, device=dev)
            touched = min(nb * hh * ww, 4 * npnt)
            add("range_to_point C=%d %dx%d" % (c, hh, ww), "pcs_range_sample_fwd_f32", timed(lambda: be.range_sample_fwd(img, fpx), reps),
                4 * c * (npnt + touched) + 12 * npnt, "bilinear gather: 4 plane reads per (point, channel), mostly L2 hits")
            be.range_sample_bwd(gp, fpx, nb, hh, ww)   # builds and caches the corner CSR
            add("range_to_point bwd C=%d %dx%d" % (c, hh, ww), "pcs_range_sample_bwd_csr_f32",
                timed(lambda: be.range_sample_bwd(gp, fpx, nb, hh, ww), reps), 4 * c * (npnt + nb * hh * ww) + 36 * npnt,
                "atomic-free; corner CSR cached on pxpy (torch's grid_sampler_2d_backward: ~15 ms per call at C=168)")
        # ---- [r5] device input pipeline: 12 raw scans -> voxelised, collated batch (hostdata.sparse_quantize_frames) -------------
        from openpcseg_amd.workloads.synthetic import device_collate, make_raw_batch
        raw = make_raw_batch(list(range(frames)))
        raw = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in raw.items()}
        npts = raw["points"].shape[0]
        mv = device_collate(raw)["lidar"].C.shape[0]
        add("device input: transform + quantize + collate, %d scans" % frames, "pcs_quantize_flags / _emit + one sort", timed(lambda: device_collate(raw), reps),
            16 * npts + 8 * npts + 12 * npts + (16 + 16 + 8) * mv, "raw points in, (voxels, features, labels) out; bound by the radix sort's passes")

    # ---- fused BatchNorm -------------------------------------------------------------------------------------
    for c in (32, 96, 256):
        rows_n = m if c < 256 else m // 8
        x = torch.randn(rows_n, c, device=dev)
        res = torch.randn(rows_n, c, device=dev)
        wgt, bias = torch.ones(c, device=dev), torch.zeros(c, device=dev)
        add("BN stats C=%d N=%d" % (c, rows_n), "pcs_bn_stats_f32", timed(lambda: be.bn_stats(x), reps), 4 * c * rows_n)
        sums = be.bn_stats(x)
        stat = be.bn_finalize(sums, float(rows_n), 1e-5, 0.1, None, None)
        add("BN apply+res+ReLU C=%d" % c, "pcs_bn_apply_f32", timed(lambda: be.bn_apply(x, res, stat, wgt, bias, True), reps),
            3 * 4 * c * rows_n)
        y, gate = be.bn_apply(x, res, stat, wgt, bias, True, want_mask=True)  # the ReLU gate as a bit mask
        dy = torch.randn(rows_n, c, device=dev)
        add("BN bwd stats C=%d" % c, "pcs_bn_bwd_stats_f32", timed(lambda: be.bn_bwd_stats(dy, x, gate, stat, True), reps),
            2 * 4 * c * rows_n, "gate read as bits")
        s2 = be.bn_bwd_stats(dy, x, gate, stat, True)
        add("BN bwd apply C=%d" % c, "pcs_bn_bwd_apply_f32",
            timed(lambda: be.bn_bwd_apply(dy, x, gate, stat, s2, float(rows_n), wgt, True, True), reps), 4 * 4 * c * rows_n,
            "gate read as bits; writes dx and dres")

    print("%d frames: %d voxels, %d points, k3 rulebook %d pairs; peak %.0f GB/s\n" % (frames, m, n, p, PEAK))
    print("| op | C-ABI entry | us / call | algorithmic MB | GB/s | of HBM peak | note |")
    print("|---|---|---|---|---|---|---|")
    for name, entry, us, mb, gbs, frac, note in rows:
        print("| %s | `%s` | %.0f | %.1f | %.0f | %.1f %% | %s |" % (name, entry, us, mb, gbs, 100 * frac, note))


if __name__ == "__main__":
    main()
