#!/usr/bin/env python3
"""Per-launch-shape durations of the kernels whose name contains a substring, from a rocprofv3 --kernel-trace CSV:
python tools/kernel_shapes.py <kernel_trace.csv> <substring> [<substring> ...]  -> (kernel, grid, workgroup): launches, avg / min / max us."""
import collections
import csv
import sys


def short(name):
    n = name.replace("void ", "").replace("(anonymous namespace)::", "")
    return n.split("(")[0][:70]


def main():
    path, subs = sys.argv[1], sys.argv[2:]
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        name = r.get("Kernel_Name", "")
        if not any(s in name for s in subs):
            continue
        grid = tuple(int(r.get("Grid_Size_" + a, r.get("Grid_Size", 0)) or 0) for a in "XYZ")
        wg = tuple(int(r.get("Workgroup_Size_" + a, r.get("Workgroup_Size", 0)) or 0) for a in "XYZ")
        us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        agg.setdefault((short(name), grid, wg), []).append(us)
    print("%-72s %-22s %-14s %6s %9s %9s %9s %10s" % ("kernel", "grid (threads)", "workgroup", "n", "avg us", "min us", "max us", "total ms"))
    for (name, grid, wg), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        print("%-72s %-22s %-14s %6d %9.1f %9.1f %9.1f %10.3f" % (name, "x".join(map(str, grid)), "x".join(map(str, wg)), len(v), sum(v) / len(v), min(v), max(v), sum(v) / 1e3))


if __name__ == "__main__":
    main()
