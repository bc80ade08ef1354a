# HBM bytes per fused-conv launch from the PMC counters (separate passes), as MI355X_MICROARCH.md prescribes:
#   bash tools/conv_traffic.sh   -> gpurun_out/conv_traffic.json (+ the raw per-kernel averages in conv_traffic_pmc.txt)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
: > $R/gpurun_out/conv_traffic_pmc.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  PCS_BENCH_PREHEAT=0 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_$c -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > /tmp/pmc_$c.log 2>&1
  f=$(find /tmp/pmc_$c -name "*counter_collection.csv" | head -1)
  cp "$f" /tmp/pmc_$c.csv
  python $R/tools/pmc_summary.py "$f" conv_os >> $R/gpurun_out/conv_traffic_pmc.txt
done
python - <<PY
import csv, json
def total(path):
    s = n = 0
    for r in csv.DictReader(open(path)):
        if "conv_os" in r.get("Kernel_Name", ""):
            s += float(r["Counter_Value"]); n += 1
    return s, n
f, nf = total("/tmp/pmc_FETCH_SIZE.csv")
w, nw = total("/tmp/pmc_WRITE_SIZE.csv")
import ctypes
lib = ctypes.CDLL("$R/openpcseg_amd/lib/libpcseg_hip.so"); lib.pcs_conv_kernel_revision.restype = ctypes.c_char_p
variants = {}
for r in csv.DictReader(open("/tmp/pmc_FETCH_SIZE.csv")):
    if "conv_os" in r.get("Kernel_Name", ""):
        variants[r["Kernel_Name"]] = variants.get(r["Kernel_Name"], 0) + 1
out = {"kernel": "conv_os5_kernel / conv_os4_kernel (all column-tile variants; fwd + dgrad launches of two bench steps: 1 warm-up + 1 timed)",
       "kernel_revision": lib.pcs_conv_kernel_revision().decode(), "variants": variants,
       "command": "bash tools/conv_traffic.sh: rocprofv3 --pmc FETCH_SIZE -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline ; same with --pmc WRITE_SIZE (separate passes)",
       "launches": nf, "fetch_size_kb_avg_raw": round(f / nf), "write_size_kb_avg": round(w / nw),
       "correction": "FETCH_SIZE doubled (gfx950 reports 1/2 of wide coalesced reads, MI355X_MICROARCH.md section HBM); WRITE_SIZE as reported",
       "hbm_bytes_per_launch": round((2 * f / nf + w / nw) * 1024)}
json.dump(out, open("$R/gpurun_out/conv_traffic.json", "w"), indent=1)
print(out)
PY
