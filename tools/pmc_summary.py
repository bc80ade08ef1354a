#!/usr/bin/env python3
"""Average the counters of a rocprofv3 --pmc CSV per kernel name. Usage: pmc_summary.py <counter_collection.csv> [name-substring]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    name = r.get("Kernel_Name", "")
    if flt and flt not in name:
        continue
    agg[name[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, cs in agg.items():
    print(name)
    for c, v in sorted(cs.items()):
        print("   %-28s n=%d avg=%.4g" % (c, len(v), sum(v) / len(v)))
