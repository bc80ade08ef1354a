#!/usr/bin/env python3
"""kernel_stats(B) - kernel_stats(A) per kernel name: two traces of the same command that differ only in the number of timed
steps leave the steady-state steps once the set-up (parameter initialisation, host-to-device copies of the model, the first
step's momentum buffers, pre-heating) cancels. python tools/stats_diff.py A.csv B.csv out.csv"""
import csv
import sys


def load(path):
    return {r["Name"]: (int(r["Calls"]), float(r["TotalDurationNs"])) for r in csv.DictReader(open(path))}


def main():
    a, b = load(sys.argv[1]), load(sys.argv[2])
    rows = []
    for name, (cb, tb) in b.items():
        ca, ta = a.get(name, (0, 0.0))
        if cb - ca > 0:
            rows.append({"Name": name, "Calls": cb - ca, "TotalDurationNs": max(tb - ta, 0.0), "AverageNs": max(tb - ta, 0.0) / (cb - ca)})
    rows.sort(key=lambda r: -r["TotalDurationNs"])
    with open(sys.argv[3], "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=["Name", "Calls", "TotalDurationNs", "AverageNs"])
        w.writeheader()
        w.writerows(rows)


if __name__ == "__main__":
    main()
