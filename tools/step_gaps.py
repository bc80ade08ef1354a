#!/usr/bin/env python3
"""GPU idle time inside the training steps of a rocprofv3 kernel trace: for the last <steps> steps (a step = the launches between
two multi-tensor optimizer bursts), the busy time (union of kernel intervals), the idle time, and the idle time charged to the
kernel that ran BEFORE each gap -- the places where the device waits for the host (size read-backs, allocator, python).
Usage: python tools/step_gaps.py <kernel_trace.csv> [steps=3] [min_gap_us=5]"""
import csv
import sys
from collections import defaultdict


def main():
    path = sys.argv[1]
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    min_gap = float(sys.argv[3]) if len(sys.argv) > 3 else 5.0
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    # step boundaries: the last optimizer kernel of every burst
    opt = [i for i, r in enumerate(rows) if "multi_tensor_apply_kernel" in r[2]]
    ends = [i for j, i in enumerate(opt) if j + 1 == len(opt) or rows[opt[j + 1]][0] - rows[i][1] > 5_000_000]
    if len(ends) < steps + 1:
        print("only %d optimizer bursts in the trace" % len(ends))
        return
    lo, hi = ends[-steps - 1] + 1, ends[-1] + 1
    seg = rows[lo:hi]
    wall = (seg[-1][1] - seg[0][0]) / 1e3
    busy_end, busy, gaps, n_gaps = seg[0][0], 0.0, defaultdict(float), defaultdict(int)
    prev_name = None
    for s, e, name in seg:
        if s > busy_end:
            g = (s - busy_end) / 1e3
            if prev_name is not None and g >= min_gap:
                gaps[(prev_name[:70], name[:70])] += g
                n_gaps[(prev_name[:70], name[:70])] += 1
            busy += (e - s) / 1e3
            busy_end = e
            prev_name = name
        elif e > busy_end:
            busy += (e - busy_end) / 1e3
            busy_end = e
            prev_name = name
    print("%d steps: wall %.2f ms/step, device busy %.2f ms/step, idle %.2f ms/step (%.1f %%), %d launches/step" %
          (steps, wall / steps / 1e3, busy / steps / 1e3, (wall - busy) / steps / 1e3, 100 * (wall - busy) / wall, len(seg) // steps))
    tot = sum(gaps.values())
    print("gaps >= %.0f us: %.2f ms/step; by (kernel before -> kernel after):" % (min_gap, tot / steps / 1e3))
    for k, v in sorted(gaps.items(), key=lambda kv: -kv[1])[:30]:
        print("  %7.1f us/step  x%-4.1f  %s  ->  %s" % (v / steps, n_gaps[k] / steps, k[0], k[1]))


if __name__ == "__main__":
    main()
