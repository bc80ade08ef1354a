mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_dense_parity.py -m gpu -q -x -k "x3 or policy" > gpurun_out/r3c9_x3_tests.log 2>&1; tail -3 gpurun_out/r3c9_x3_tests.log
cat > /tmp/x3_sweep.py <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from openpcseg_amd import functional as F, native
from openpcseg_amd.workloads.synthetic import make_batch
dev = torch.device("cuda:0")
coords = make_batch(list(range(12)))["lidar"].C.to(dev)
coords = coords[torch.argsort(F.sphash(coords))].contiguous()
levels, ts = [coords], 1
for _ in range(4):
    levels.append(F.spdownsample(levels[-1], 2, 2, ts)); ts *= 2
be = native.backend()
def timed(run, reps=40, warm=60):
    for _ in range(warm): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
for level, cin, cout in [(0,96,96),(0,32,32),(1,96,96),(1,64,64),(2,128,128),(2,192,128),(2,64,64),(3,256,256),(3,384,256),(3,128,128),(4,256,256),(1,112,112),(3,672,448)]:
    c = levels[level]
    entry = F.build_kernel_map(c, c, (3,3,3), (2**level,)*3, (1,1,1))
    n, p = c.shape[0], entry.fwd.num_pairs
    x = torch.randn(n, cin, device=dev); w = torch.randn(27, cin, cout, device=dev) * 0.05
    wp = be.prepare_weights_x3(w, transpose=False)
    t32 = timed(lambda: be.conv_gather_gemm(x, w, entry.fwd))
    res = []
    for tile in (None, 256, 384):
        try:
            tx = timed(lambda: be.conv_gather_gemm_x3(x, wp, 27, cout, entry.fwd, tile_rows=tile))
            res.append("%s:%.0f" % (tile, tx))
        except Exception as e:
            res.append("%s:err" % tile)
    fl = 2.0 * p * cin * cout
    best = min(float(r.split(":")[1]) for r in res if not r.endswith("err"))
    print("level=%d %d->%d fp32 %.0f us (%.1f TF)  x3 [%s] us  best %.1f TF  speed-up %.2fx" % (level, cin, cout, t32, fl/t32/1e6, " ".join(res), fl/best/1e6, t32/best), flush=True)
PY

: > gpurun_out/r3c9_x3_R.txt
for r in 2 3 4; do echo "== PCS_CONVX_R=$r" >> gpurun_out/r3c9_x3_R.txt; PCS_CONVX_R=$r timeout 300 python /tmp/x3_sweep.py >> gpurun_out/r3c9_x3_R.txt 2>&1; done
grep -v amdgpu gpurun_out/r3c9_x3_R.txt | cut -c1-150
