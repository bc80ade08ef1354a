# HBM bytes per fused-conv launch from the PMC counters (separate passes), as MI355X_MICROARCH.md prescribes:
#   bash tools/conv_traffic.sh   -> gpurun_out/conv_traffic.json (+ the raw per-kernel averages in conv_traffic_pmc.txt)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
: > $R/gpurun_out/conv_traffic_pmc.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  PCS_BENCH_PREHEAT=0 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_$c -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-split-line --no-device-input-line --models none > /tmp/pmc_$c.log 2>&1
  f=$(find /tmp/pmc_$c -name "*counter_collection.csv" | head -1)
  cp "$f" /tmp/pmc_$c.csv
  python $R/tools/pmc_summary.py "$f" conv_os >> $R/gpurun_out/conv_traffic_pmc.txt
done
python - <<PY
import csv, json, ctypes
lib = ctypes.CDLL("$R/openpcseg_amd/lib/libpcseg_hip.so"); lib.pcs_conv_kernel_revision.restype = ctypes.c_char_p
def total(path, pred):
    s = n = 0
    for r in csv.DictReader(open(path)):
        if pred(r.get("Kernel_Name", "")):
            s += float(r["Counter_Value"]); n += 1
    return s, n
out = {"kernel_revision": lib.pcs_conv_kernel_revision().decode(),
       "command": "bash tools/conv_traffic.sh: rocprofv3 --pmc FETCH_SIZE -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline ; same with "
                  "--pmc WRITE_SIZE (separate passes); the default bench run holds the fp32 step and the bf16 step",
       "correction": "FETCH_SIZE doubled (gfx950 reports 1/2 of wide coalesced reads, MI355X_MICROARCH.md section HBM); WRITE_SIZE as reported"}
for key, pred in (("f32", lambda n: "conv_os" in n and "conv_os5h" not in n and "conv_os6h" not in n and "conv_os5x" not in n), ("half", lambda n: "conv_os5h" in n or "conv_os6h" in n)):
    f, nf = total("/tmp/pmc_FETCH_SIZE.csv", pred)
    w, nw = total("/tmp/pmc_WRITE_SIZE.csv", pred)
    if not nf or not nw:
        continue
    variants = {}
    for r in csv.DictReader(open("/tmp/pmc_FETCH_SIZE.csv")):
        if pred(r.get("Kernel_Name", "")):
            variants[r["Kernel_Name"]] = variants.get(r["Kernel_Name"], 0) + 1
    out[key] = {"launches": nf, "variants": variants, "fetch_size_kb_avg_raw": round(f / nf), "write_size_kb_avg": round(w / nw),
                "hbm_bytes_per_launch": round((2 * f / nf + w / nw) * 1024)}
json.dump(out, open("$R/gpurun_out/conv_traffic.json", "w"), indent=1)
print(out)
PY
