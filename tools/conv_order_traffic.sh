#!/bin/bash
# HBM read traffic (rocprofv3 --pmc FETCH_SIZE, raw KB as reported; x2 per the gfx950 note for bytes) of the fused convolution under
# the launch orders of tools/conv_xcd_order_ab.py, one process per order:  bash tools/conv_order_traffic.sh "<level> <cin> <cout> [bf16]" ...
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for o in ${PCS_ORDERS:-rows heavy xcd xcdheavy xb16 xb32 xb64}; do
  rm -rf /tmp/pmc_o_$o
  PCS_AB_ONLY=$o PCS_SWEEP_REPS=4 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_o_$o -- python $R/tools/conv_xcd_order_ab.py "$@" > /tmp/pmc_o_$o.log 2>&1
  f=$(find /tmp/pmc_o_$o -name "*counter_collection.csv" | head -1)
  echo "== order $o"
  python - "$f" <<'PY'
import csv, sys
from collections import defaultdict
s, n = defaultdict(float), defaultdict(int)
for r in csv.DictReader(open(sys.argv[1])):
    k = r.get("Kernel_Name", "")
    if "conv_os" in k:
        key = (k[k.index("conv_os"):][:40], r.get("Grid_Size", ""))
        s[key] += float(r["Counter_Value"]); n[key] += 1
for k in sorted(s):
    print("  %-62s grid %-10s launches %3d  FETCH_SIZE avg %9.0f KB (x2 = %7.1f MB)" % (k[0], k[1], n[k], s[k] / n[k], 2 * s[k] / n[k] / 1024))
PY
done
