#!/usr/bin/env python3
"""Per-step kernel budget by owner from a `rocprofv3 --kernel-trace --stats` kernel_stats.csv.
Usage: python tools/step_budget.py <kernel_stats.csv> <steps in the trace> [title]   -> markdown on stdout
(`tools/profile_bench.sh` traces 3 steps: 1 warm-up + 2 timed, no pre-heat.)"""
import collections
import csv
import sys

OWNERS = (
    ("conv fwd / dgrad (fused gather-GEMM-scatter)", ("conv_os", "conv_ring")),
    ("conv wgrad", ("wgrad",)),
    ("weight preparation (half / split planes, transposes)", ("prepare_weights", "transpose_weights", "transpose_kab")),
    ("BatchNorm (partials, reduce, finalize, apply, backward)", ("bn_",)),
    ("kernel maps (hash tables, probes, segments, downsample)", ("rb_", "table_", "kmap", "downsample", "hash_", "coord", "unique", "pairs_", "ds_pack", "ds_unpack")),
    ("point <-> voxel (voxelize, devoxelize, trilinear corners)", ("voxelize", "corner_map", "ti_weight", "count_kernel", "quantize")),
    ("Lovasz / sorts (rocPRIM)", ("rocprim", "lovasz")),
    ("optimizer (multi-tensor SGD)", ("multi_tensor_apply",)),
    ("dense GEMMs (point MLPs, classifier: rocBLAS / hipBLASLt)", ("Cijk_",)),
    ("copies / fills (runtime)", ("__amd_rocclr_",)),
    ("torch elementwise / reductions / indexing", ("at::native", "at_cuda_detail", "at::cuda", "softmax_warp")),
)


def owner_of(name):
    for owner, keys in OWNERS:
        if any(k in name for k in keys):
            return owner
    return "other"


def main():
    path, steps = sys.argv[1], int(sys.argv[2])
    title = sys.argv[3] if len(sys.argv) > 3 else path
    rows = list(csv.DictReader(open(path)))
    agg = collections.OrderedDict((o, [0, 0.0, collections.Counter()]) for o, _ in OWNERS)
    agg["other"] = [0, 0.0, collections.Counter()]
    for r in rows:
        a = agg[owner_of(r["Name"])]
        a[0] += int(r["Calls"])
        a[1] += float(r["TotalDurationNs"])
        short = r["Name"].replace("void ", "").replace("(anonymous namespace)::", "").replace("rocprim::ROCPRIM_400001_NS::detail::", "rocprim::")
        a[2][short.split("(")[0].split("<")[0][:48]] += float(r["TotalDurationNs"])
    tot_calls = sum(a[0] for a in agg.values())
    tot_ns = sum(a[1] for a in agg.values())
    print("### %s\n" % title)
    print("| owner | launches / step | ms / step | share | largest kernels (ms / step) |")
    print("|---|---|---|---|---|")
    for owner, (calls, ns, names) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        if not calls:
            continue
        top = ", ".join("`%s` %.2f" % (n, v / steps / 1e6) for n, v in names.most_common(3))
        print("| %s | %.0f | %.2f | %.1f %% | %s |" % (owner, calls / steps, ns / steps / 1e6, 100 * ns / tot_ns, top))
    print("| **total** | **%.0f** | **%.2f** | | |" % (tot_calls / steps, tot_ns / steps / 1e6))


if __name__ == "__main__":
    main()
