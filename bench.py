#!/usr/bin/env python3
"""bench.py -- LiDAR frames/sec, MinkUNet-34 cr1.0 training step at SemanticKITTI shape.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = one pass of the hot path over one batch: forward (hash-grid voxelise, 9 rulebook
builds, 56 fused sparse convs, 4 trilinear devoxelisations), CE + Lovasz loss, backward
(dgrad + wgrad per conv), gradient all-reduce (N > 1, RCCL), grad clip, SGD step. Inputs are
synthetic ~120k-point scans (openpcseg_amd/workloads/synthetic.py), voxelised on the host by
the reference's dataset transform BEFORE the timed region and resident in HBM; rulebooks are
rebuilt every step (no cross-iteration cache, as in the reference).
Prints ONE JSON line (rank 0): `value` = the fp32 step (fp32 MFMA convolutions; weight gradient policy in
`config.wgrad`) with the roofline of the dominant kernel (fused gather-GEMM conv) measured live with HIP
events on the launch stream; `amp_bf16` = the same step under torch.autocast(bf16) (the reference trains
under --amp); `fp32_bf16x3` = the fp32 step with the forward / input-gradient convolutions on the
three-plane split kernel (fp32 in / out on the bf16 MFMAs, opt-in, fp32-grade); `comm` (N > 1) = RCCL time
per step and its overlap with backward from a device trace; `cpu_baseline` = the reference's own compiled
CPU backend on ONE full frame of the workload (N = 1 only, baseline only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from openpcseg_amd import native  # noqa: E402
from openpcseg_amd.sparse import SparseTensor  # noqa: E402
from openpcseg_amd.workloads.minkunet import MK34_LAYERS, MinkUNet  # noqa: E402
from openpcseg_amd.workloads.synthetic import make_batch  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 / 32x32x2 dense peak
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 / fp16 MFMA peak
PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E spec peak (6.3 TB/s achievable)
TRAFFIC_FILE = "round6_conv_traffic.json"
# what the counters say binds conv_os5h_kernel (profiles/round3_convh_pmc.md); filled in from the PMC passes of round 3
HALF_BINDING_NOTE = ("gather latency / vector L1, not HBM and not the MFMA pipe [r6 counters, profiles/round6_convh_ws.md]: MFMA pipe 15 % busy, "
                     "TA 63 %, vector L1 ~60 % of 64 B/clk/CU, waves parked 47 %; conv_os6h keeps the weights stationary in registers and "
                     "prefetches the gathered rows a sub-group ahead (round 5's conv_os5h: TA 65-81 %, MFMA 10-19 %)")
POINTS_PER_FRAME = 120000


class ConvMeter:
    """Wraps the backend's fused conv launches (fp32: conv_gather_gemm, half: conv_gather_gemm_h): HIP events on the
    launch stream (torch's current stream IS the stream the C ABI launches on) + the algorithmic work per launch,
    2*P*Cin*Cout flop and e*(Nin*Cin + Nout*Cout) + 8P + e*K*Cin*Cout bytes."""

    def __init__(self, be):
        self.be = be
        self.orig = {"f32": be.conv_gather_gemm, "half": be.conv_gather_gemm_h, "x3": be.conv_gather_gemm_x3}
        self.records, self.enabled = [], False

    def __enter__(self):
        def timed(call, kind, kmap, k, cin, cout, n_src):
            if not self.enabled:
                return call()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = call()
            e1.record()
            self.records.append((e0, e1, kmap, (k, cin, cout), n_src, kind))  # pair counts are read at summary time
            return out

        def wrapped32(src, weight, kmap, bias=None, tile_rows=None, **kw):
            k, cin, cout = weight.shape
            return timed(lambda: self.orig["f32"](src, weight, kmap, bias, tile_rows, **kw), "f32", kmap, k, cin, cout,
                         src.shape[0])

        def wrapped16(src, wp, k, cout, kmap, bias=None, tile_rows=None, **kw):
            return timed(lambda: self.orig["half"](src, wp, k, cout, kmap, bias, tile_rows, **kw), "half", kmap, k,
                         src.shape[1], cout, src.shape[0])

        def wrappedx3(src, wp, k, cout, kmap, bias=None, tile_rows=None, **kw):
            return timed(lambda: self.orig["x3"](src, wp, k, cout, kmap, bias, tile_rows, **kw), "x3", kmap, k,
                         src.shape[1], cout, src.shape[0])
        self.be.conv_gather_gemm, self.be.conv_gather_gemm_h, self.be.conv_gather_gemm_x3 = wrapped32, wrapped16, wrappedx3
        return self

    def __exit__(self, *a):
        self.be.conv_gather_gemm, self.be.conv_gather_gemm_h = self.orig["f32"], self.orig["half"]
        self.be.conv_gather_gemm_x3 = self.orig["x3"]

    def summary(self, amp, split=False):
        kind = "half" if amp else "f32"
        recs = [r for r in self.records if r[5] == kind or (split and not amp and r[5] == "x3")]
        if not recs:
            return None
        e = 2.0 if amp else 4.0
        ms = sum(r[0].elapsed_time(r[1]) for r in recs)
        flops = abytes = 0.0
        for _, _, kmap, (k, cin, cout), n_src, _ in recs:
            p = kmap.num_pairs
            flops += 2.0 * p * cin * cout
            abytes += e * (n_src * cin + kmap.n_dst * cout) + 8.0 * p + e * k * cin * cout
        n = len(recs)
        tflops = flops / (ms * 1e-3) / 1e12
        common = {"launches": n, "avg_launch_us": round(ms * 1e3 / n, 2), "flops_per_launch": round(flops / n),
                  "algorithmic_bytes_per_launch": round(abytes / n)}
        if split and not amp:
            # fp32 operands as three bf16 planes, six plane products per algorithmic product on the 16-bit MFMA pipe: its
            # roof for ALGORITHMIC flops is the dense bf16 peak / 6 (the thin layers the split kernel does not serve stay
            # on the fp32 MFMA kernel and are part of the same average)
            nx = sum(1 for r in recs if r[5] == "x3")
            peak = PEAK_BF16_MFMA_TFLOPS / 6.0
            return dict({"kernel": "conv_os5x_kernel (pcs_conv_gather_gemm_f32_bf16x3: fwd + dgrad; %d of %d launches, the rest "
                                   "on conv_os5_kernel)" % (nx, n), "bound": "mfma", "achieved": round(tflops, 3),
                         "peak": round(peak, 1), "unit": "TFLOP/s", "frac": round(tflops / peak, 4),
                         "peak_note": "dense bf16 MFMA peak / 6 plane products per algorithmic product",
                         "vs_fp32_mfma_peak": round(tflops / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": None}, **common)
        # HBM bytes per launch come from separate rocprofv3 --pmc passes (tools/conv_traffic.sh; rocprofv3 cannot run
        # inside bench.py). The file names the kernel revision it was measured on; a stale file is refused.
        traffic, note = None, "no PMC traffic file for this kernel revision (run tools/conv_traffic.sh)"
        tfile = os.path.join(ROOT, "profiles", TRAFFIC_FILE)
        rev = native.load_library().pcs_conv_kernel_revision().decode()
        if os.path.exists(tfile):
            tj = json.load(open(tfile))
            ent = tj.get("half" if amp else "f32")
            if tj.get("kernel_revision") == rev and ent:
                traffic = ent.get("hbm_bytes_per_launch")
                note = ("HBM bytes/launch, rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE (separate passes, FETCH doubled per the "
                        "gfx950 note), profiles/%s, kernel revision %s" % (TRAFFIC_FILE, rev))
            else:
                note = "profiles/%s was measured on kernel revision %s, this build is %s: refused" % (
                    TRAFFIC_FILE, tj.get("kernel_revision"), rev)
        if amp:
            # 16-bit MFMA is 16x the fp32 rate: the fused conv is priced on its ALGORITHMIC bytes (each feature row once,
            # weights once, rulebook once) against HBM; mfma_tflops is the same launches against the MFMA roof
            gbs = abytes / (ms * 1e-3) / 1e9
            return dict({"kernel": "conv_os6h_kernel + conv_os5h_kernel (pcs_conv_gather_gemm_h: fwd + dgrad, %s)" % amp, "bound": "hbm",
                         "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4),
                         "traffic": traffic, "traffic_note": note, "mfma_tflops": round(tflops, 1),
                         "mfma_frac_of_dense_bf16_peak": round(tflops / PEAK_BF16_MFMA_TFLOPS, 4),
                         "binding_unit": HALF_BINDING_NOTE}, **common)
        return dict({"kernel": "conv_os5_kernel / conv_os4_kernel (pcs_conv_gather_gemm_f32: fwd + dgrad)", "bound": "mfma",
                     "achieved": round(tflops, 3), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(tflops / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": traffic, "traffic_note": note}, **common)


class ClockSampler:
    """Shader clock / socket power of the GPU during the timed region, sampled from a host thread (amdsmi, else
    sysfs hwmon): the fp32-MFMA kernels scale with the sustained clock (boxes of this pool differ by up to 1.4x on
    them while the HBM-bound kernels agree), so the bench line carries the clock it was measured at."""

    def __init__(self, dev_index, period_s=0.1):
        import threading
        self.period, self.samples, self.src = period_s, [], None
        self.xcd_min, self.hotspot, self._throttle, self._thr0 = [], [], None, {}
        self._stop = threading.Event()
        self._read = self._probe(dev_index)
        self._thr = threading.Thread(target=self._run, daemon=True) if self._read else None

    def _probe(self, dev_index):
        try:
            import amdsmi
            amdsmi.amdsmi_init()
            h = amdsmi.amdsmi_get_processor_handles()[dev_index]
            ct = getattr(amdsmi.AmdSmiClkType, "GFX", None) or amdsmi.AmdSmiClkType.SYS

            def metrics():
                try:
                    return amdsmi.amdsmi_get_gpu_metrics_info(h)
                except Exception:
                    return {}

            def read():
                clk = amdsmi.amdsmi_get_clock_info(h, ct).get("clk")
                pw = amdsmi.amdsmi_get_power_info(h)
                p = pw.get("current_socket_power")
                if not isinstance(p, (int, float)):
                    p = pw.get("average_socket_power")
                m = metrics()
                xcd = [float(v) for v in (m.get("current_gfxclks") or []) if isinstance(v, (int, float)) and v > 0]
                if xcd:  # the slowest XCD bounds a launch that spans all eight
                    self.xcd_min.append(min(xcd))
                t = m.get("temperature_hotspot")
                if isinstance(t, (int, float)):
                    self.hotspot.append(float(t))
                return (float(clk) if isinstance(clk, (int, float)) else None,
                        float(p) if isinstance(p, (int, float)) else None)

            def throttle():
                m = metrics()
                return {k: m[k] for k in ("ppt_residency_acc", "prochot_residency_acc", "socket_thm_residency_acc",
                                          "vr_thm_residency_acc", "hbm_thm_residency_acc", "energy_accumulator")
                        if isinstance(m.get(k), (int, float))}
            read()
            self._throttle = throttle
            self.src = "amdsmi"
            return read
        except Exception:
            pass
        try:
            import glob
            cards = sorted(glob.glob("/sys/class/drm/card[0-9]*/device/hwmon/hwmon*/freq1_input"))
            f = cards[dev_index]
            pw = os.path.join(os.path.dirname(f), "power1_average")

            def read():
                clk = float(open(f).read()) / 1e6
                p = float(open(pw).read()) / 1e6 if os.path.exists(pw) else None
                return clk, p
            read()
            self.src = "sysfs"
            return read
        except Exception:
            return None

    def _run(self):
        while not self._stop.is_set():
            try:
                self.samples.append(self._read())
            except Exception:
                pass
            self._stop.wait(self.period)

    def __enter__(self):
        if self._throttle:
            self.xcd_min, self.hotspot = [], []
            try:
                self._thr0 = self._throttle()
            except Exception:
                self._thr0 = {}
        if self._thr:
            self._thr.start()
        return self

    def __exit__(self, *a):
        self._stop.set()
        if self._thr:
            self._thr.join(timeout=2)

    def summary(self):
        clk = [c for c, _ in self.samples if c]
        pw = [p for _, p in self.samples if p]
        if not clk:
            return {"sclk_mhz_mean": None, "source": self.src, "note": "no clock source readable on this host"}
        out = {"sclk_mhz_mean": round(sum(clk) / len(clk), 1), "sclk_mhz_min": min(clk), "sclk_mhz_max": max(clk),
               "samples": len(clk), "source": self.src}
        if pw:
            out["socket_power_w_mean"] = round(sum(pw) / len(pw), 1)
        if self.xcd_min:
            out["slowest_xcd_mhz_mean"] = round(sum(self.xcd_min) / len(self.xcd_min), 1)
            out["slowest_xcd_mhz_min"] = min(self.xcd_min)
        if self.hotspot:
            out["hotspot_c_max"] = max(self.hotspot)
        if self._throttle and self._thr0:
            try:  # throttler residency counters over the timed region (power / thermal limits acting on the clocks)
                t1 = self._throttle()
                out["throttle_residency_delta"] = {k.replace("_residency_acc", ""): t1[k] - self._thr0[k]
                                                   for k in t1 if k in self._thr0 and k != "energy_accumulator"}
            except Exception:
                pass
        return out


def to_device(batch, dev):
    lidar, tg = batch["lidar"], batch["targets"]
    coords = lidar.C.to(dev)
    return {"lidar": SparseTensor(lidar.F.to(dev), coords), "targets": SparseTensor(tg.F.to(dev), coords)}


def fresh(b):
    """New containers over the SAME device tensors: every step rebuilds all maps."""
    return {"lidar": SparseTensor(b["lidar"].F, b["lidar"].C), "targets": SparseTensor(b["targets"].F, b["targets"].C)}


CPU_BASELINE_THREADS = 8      # the reference's CPU code forks an OpenMP team per row: more threads = slower
CPU_BASELINE_TIMEOUT_S = 900  # hard cap (one full frame took 226.5 s on the build container's 8 cores)


def _cpu_baseline_worker():
    """Runs in a subprocess (OMP_NUM_THREADS fixed before any OpenMP runtime starts): the reference's own compiled CPU
    backend (oracle/_ref) under the same MinkUNet-34 cr1.0 training step (fwd + loss + bwd) on ONE FULL frame of the
    bench workload (seed 0, 120 000 rays), timed once after a 500-ray warm-up. Rounds 1-2 extrapolated from sub-sampled
    frames; the cost per ray is far from constant on this backend (3.9 -> 2.9 -> 1.9 ms/ray from 6 k to 20 k to 120 k
    rays: per-offset matmuls of a few hundred rows at the small sizes), so any fit over a bounded sub-sample costs as
    much as the frame itself or is off by 2-3x. One frame = the bounded sample (1/12 of one GPU step's batch)."""
    torch.set_num_threads(CPU_BASELINE_THREADS)
    try:
        from oracle.adapter import RefBackend
        ref, kind = RefBackend(), "reference"
    except Exception:
        from oracle.adapter import OracleBackend
        ref, kind = OracleBackend(), "port"
    native._BACKEND = ref  # cpu_baseline leg only: the thing timed here IS the CPU reference
    torch.manual_seed(0)
    model = MinkUNet(num_class=20, num_layer=MK34_LAYERS, cr=1.0).train()

    def run(n_points):
        b = make_batch([0], n_points=n_points)
        batch = {"lidar": SparseTensor(b["lidar"].F, b["lidar"].C), "targets": SparseTensor(b["targets"].F, b["targets"].C)}
        t0 = time.perf_counter()
        out = model(batch)
        t1 = time.perf_counter()
        out["loss"].backward()
        t2 = time.perf_counter()
        model.zero_grad(set_to_none=True)
        return t1 - t0, t2 - t1, b["lidar"].C.shape[0]

    n_rays = int(os.environ.get("PCS_CPU_BASELINE_RAYS", POINTS_PER_FRAME))  # test rigs shrink the frame
    run(500)  # warm-up: pages the backend and the OpenMP runtime in
    fwd, bwd, n_vox = run(n_rays if n_rays < POINTS_PER_FRAME else None)
    cores = CPU_BASELINE_THREADS if kind == "reference" else 1
    print(json.dumps({"value": round(1.0 / (fwd + bwd), 5), "unit": "frames/s", "cores": cores, "kind": kind,
                      "seconds_per_frame": round(fwd + bwd, 1),
                      "sample": "n=1: one full frame (%d rays, %d voxels) MinkUNet-34 fwd %.0f s + bwd %.0f s after a 500-ray warm-up, no extrapolation"
                                % (n_rays, n_vox, fwd, bwd)}),
          flush=True)


def _cpu_pytorch_worker():
    """Runs in a subprocess: BASELINE config 1 -- the package's pure-PyTorch CPU path (openpcseg_amd/cpu_fallback.py: gather /
    index_add_ / searchsorted forms of every op, TS:torchsparse/nn/functional/conv.py:67-79 semantics) under the reference's own
    SPVCNN source (point branch + voxelize / devoxelize; the fused MinkUNet-18 workload where the reference sources are not
    staged), one 2 000-point synthetic scan, fwd + loss + bwd, one warm-up + the median of three runs."""
    import statistics
    torch.set_num_threads(CPU_BASELINE_THREADS)
    from openpcseg_amd import cpu_fallback
    cpu_fallback.install()
    torch.Tensor.cuda = lambda self, *a, **k: self   # the reference's model files call .cuda() on their targets (minkunet.py:425)
    b = make_batch([0], n_points=2000)
    model, what = None, "fused MinkUNet-18 workload"
    try:
        sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
        import fullsize as fs
        import make_golden as mg
        import openpcseg_amd
        openpcseg_amd.install_reference_aliases()
        dotted, cls = fs.MODEL_PATH["config3"]
        model = getattr(mg.import_reference_model(dotted), cls)(mg._AttrDict(fs.MODEL_CFG["config3"]), 20)
        what = "the reference's SPVCNN mk18 cr1.0 source"
    except Exception:
        from openpcseg_amd.workloads.minkunet import MK18_LAYERS
        model = MinkUNet(num_class=20, num_layer=MK18_LAYERS, cr=1.0)
    torch.manual_seed(0)
    model.train()

    def run():
        batch = {"lidar": SparseTensor(b["lidar"].F, b["lidar"].C), "targets": SparseTensor(b["targets"].F, b["targets"].C),
                 "offset": None}
        t0 = time.perf_counter()
        out = model(batch)
        out = out[0] if isinstance(out, tuple) else out
        out["loss"].backward()
        model.zero_grad(set_to_none=True)
        return time.perf_counter() - t0
    run()
    ts = [run() for _ in range(3)]
    med = statistics.median(ts)
    print(json.dumps({"value": round(1.0 / med, 4), "unit": "frames/s", "cores": CPU_BASELINE_THREADS, "kind": "pytorch",
                      "seconds_per_frame": round(med, 3),
                      "sample": "config 1: %s on the pure-PyTorch CPU path, one 2000-pt scan (%d voxels), fwd+bwd, median of 3 after 1 warm-up"
                                % (what, b["lidar"].C.shape[0])}), flush=True)


def _run_worker(flag, timeout_s):
    import subprocess
    env = dict(os.environ, OMP_NUM_THREADS=str(CPU_BASELINE_THREADS), MKL_NUM_THREADS=str(CPU_BASELINE_THREADS),
               HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), flag, "1"], env=env, capture_output=True, text=True,
                           timeout=timeout_s)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if line:
            return json.loads(line[-1])
        return {"value": None, "sample": "worker failed: " + (r.stderr.strip().splitlines() or ["?"])[-1][:160]}
    except subprocess.TimeoutExpired:
        return {"value": None, "sample": "worker exceeded the %d s cap on this host" % timeout_s}


def cpu_baseline(compact=True):
    """`cpu_baseline` of the bench line: the reference's compiled CPU backend on one full frame of the workload, and beside it
    (`pytorch`) BASELINE config 1 on the package's pure-PyTorch CPU path."""
    ref = _run_worker("--cpu-baseline-worker", CPU_BASELINE_TIMEOUT_S)
    ref.setdefault("unit", "frames/s")
    ref.setdefault("cores", CPU_BASELINE_THREADS)
    ref.setdefault("kind", "reference")
    pt = _run_worker("--cpu-pytorch-worker", 300)
    if compact:   # the driver keeps the last 2 000 characters of the line: short sample texts (the long ones: --verbose-json)
        ref.pop("sample_long", None)
        if ref.get("value") is not None:
            ref["sample"] = "n=1 full frame fwd+bwd, reference CPU backend (oracle/_ref), no extrapolation"
        pt = {k: pt.get(k) for k in ("value", "seconds_per_frame", "sample") if k in pt}
        if pt.get("value") is not None:
            pt["sample"] = "config 1: reference SPVCNN, one 2000-pt scan, pure-PyTorch CPU path"
    ref["pytorch"] = pt
    return ref


LINE_LIMIT = 1950   # the driver keeps the last 2 000 characters of the line: a longer one would lose its HEAD (metric, value)


def fit_line(res, limit=LINE_LIMIT):
    """The compact JSON line, trimmed to `limit` characters if a record grew unexpectedly (a long worker error text, more model rows):
    text first (labels, samples, notes), then whole secondary records from the least to the most important; the headline keys, `roofline`
    and `cpu_baseline`'s numbers are never touched. Returns the line; `res` is trimmed in place."""
    def line():
        return json.dumps(res, separators=(",", ":"))

    def drop(path):
        d = res
        for k in path[:-1]:
            d = d.get(k) if isinstance(d, dict) else None
            if d is None:
                return
        if isinstance(d, dict):
            d.pop(path[-1], None)
    for path in (("models", "fmt"), ("cpu_baseline", "pytorch", "sample"), ("cpu_baseline", "sample"), ("config", "notes"),
                 ("models", "minkunet18/fused"), ("cpu_baseline", "pytorch"), ("fp32_bf16x3",), ("models",), ("device_input",),
                 ("amp_bf16", "roofline"), ("comm",), ("amp_bf16",)):
        if len(line()) <= limit:
            break
        drop(path)
        res["trimmed"] = True
    return line()


_PREHEATED = {"done": False}


def preheat(step, distributed, dev, window=10, max_windows=40):
    """Untimed clock warm-up before the W warm-up steps. Some MI355X boxes start a fresh process well below their
    sustained clocks and take tens of seconds of load to get there (measured: the same binary at 173 -> 152 -> 137 ms
    per step over its first minute on one box -- about 0.5 % per second --, flat at 137 ms from the first step on
    others). Windows of `window` steps are run until a window is no longer > 0.4 % faster than the one before (at
    least three windows, at most `max_windows`); with several ranks the decision is shared so that every rank runs the
    same number of steps.
    The K timed steps that follow are full, unmodified steps. PCS_BENCH_PREHEAT=0 skips this.
    [r5] The FIRST record measured in a process additionally holds the load for at least PCS_BENCH_PREHEAT_MIN_S seconds (default 60):
    boxes of the pool ran the first 10 ... 40 s of a fresh process at a constant 1.08-1.2x slower rate per launch (a plateau, not a
    ramp: three equal windows ended the pre-heat after 4 s) and every later record of the same process at full speed
    (profiles/round5_bench_first_record_slow_state.log, round5_bench_second_box_long_slow_state.log)."""
    if os.environ.get("PCS_BENCH_PREHEAT", "1") == "0":
        return
    min_s = 0.0 if _PREHEATED["done"] else float(os.environ.get("PCS_BENCH_PREHEAT_MIN_S", "60"))
    _PREHEATED["done"] = True
    t_start = time.perf_counter()
    prev, wi = None, 0
    while True:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(window):
            step()
        torch.cuda.synchronize()
        cur = (time.perf_counter() - t0) / window
        if os.environ.get("PCS_BENCH_PREHEAT_LOG") == "1":
            print("preheat window: %.1f ms/step" % (cur * 1e3), file=sys.stderr, flush=True)
        held = time.perf_counter() - t_start >= min_s          # the minimum hold counts in seconds, not in windows
        improving = wi < 2 or cur < 0.996 * prev
        go = 1 if (not held or (improving and wi + 1 < max_windows)) else 0
        if distributed:
            flag = torch.tensor([go], device=dev, dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            go = int(flag.item())
        prev, wi = cur, wi + 1
        if not go:
            break


def comm_profile(step, rank):
    """One extra, untimed step under torch.profiler (N > 1 only): RCCL kernel time per step and how much of it ran
    beside compute kernels -- DDP's bucketed gradient all-reduce is supposed to overlap the rest of backward
    (R:train.py:215-219 wraps the model in DistributedDataParallel). Every rank runs the step; rank 0 reports."""
    if rank != 0:  # every rank runs the step (its collectives); only rank 0 traces it
        step()
        torch.cuda.synchronize()
        return None
    try:
        from torch.profiler import ProfilerActivity, profile
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            step()
            torch.cuda.synchronize()
        return comm_overlap_summary(prof.events())
    except Exception as e:  # measurement aid only: never fatal for the bench line
        return {"error": str(e)[:200]}


def comm_overlap_summary(events):
    """RCCL kernels vs everything else on the device timeline of one step (microsecond intervals from the profiler)."""
    comm, comp = [], []
    for ev in events:
        if getattr(ev, "device_type", None) is None or "cuda" not in str(ev.device_type).lower():
            continue
        t0 = ev.time_range.start
        t1 = ev.time_range.end
        if t1 <= t0:
            continue
        name = ev.name.lower()
        (comm if ("nccl" in name or "rccl" in name) else comp).append((t0, t1, ev.name))
    if not comm:
        return {"rccl_kernels": 0, "note": "no RCCL kernel in the step's device trace"}

    def union(iv):
        iv = sorted((a, b) for a, b, _ in iv)
        out = []
        for a, b in iv:
            if out and a <= out[-1][1]:
                out[-1][1] = max(out[-1][1], b)
            else:
                out.append([a, b])
        return out
    cu, pu = union(comm), union(comp)
    total = sum(b - a for a, b in cu)
    overl = 0.0
    for a, b in cu:
        for c, d in pu:
            if d <= a or c >= b:
                continue
            overl += min(b, d) - max(a, c)
    convs = [iv for iv in comp if "conv_os" in iv[2] or "wgrad" in iv[2]]
    last_conv_end = max(b for _, b, _ in convs) if convs else None
    first_comm = min(a for a, _, _ in comm)
    return {"rccl_kernels": len(comm), "rccl_ms_per_step": round(total / 1e3, 3),
            "overlapped_with_compute_ms": round(overl / 1e3, 3), "exposed_ms": round((total - overl) / 1e3, 3),
            "first_rccl_kernel_before_last_conv_ends": (None if last_conv_end is None else bool(first_comm < last_conv_end)),
            "first_rccl_to_last_conv_end_ms": (None if last_conv_end is None else round((last_conv_end - first_comm) / 1e3, 3))}


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks ourselves, exactly as the driver
    would (R:dist_train.sh:17-19 -> torch.distributed.launch; one process per GPU, rendezvous on 127.0.0.1), and hand
    the exit code on. Under a launcher (WORLD_SIZE set) this is a no-op; main() then checks WORLD_SIZE == --gpus."""
    if "WORLD_SIZE" in os.environ or args.gpus <= 1:
        return
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames-per-gpu", type=int, default=12)  # BATCH_SIZE_PER_GPU of minkunet_mk34_cr10.yaml
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--amp", choices=["off", "bf16", "fp16"], default="off",
                    help="mixed precision like the reference's --amp (second bench line; the headline metric is fp32)")
    ap.add_argument("--no-amp-line", action="store_true", help="skip the secondary bf16 record of the default run")
    ap.add_argument("--no-split-line", action="store_true", help="skip the fp32_bf16x3 record (split-kernel convolutions) of the default run")
    ap.add_argument("--wgrad", choices=["fp32", "bf16x3"], default="fp32",
                    help="fp32 weight gradient of the >= 96-channel convolutions: fp32 MFMA, or fp32 operands as three bf16 planes "
                         "on the 16-bit MFMAs (fp32-grade: six plane products, error vs float64 <= 2x the fp32 MFMA path's)")
    ap.add_argument("--cpu-baseline-worker", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--cpu-pytorch-worker", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--models", default="auto",
                    help="secondary `models` record: comma list of {minkunet18,spvcnn18,cylinder,rpvnet34,minkunet34}[:reference|fuse|workload], "
                         "'auto' (N = 1 default run: the reference's segmentor sources on the HIP backend, unmodified and after "
                         "openpcseg_amd.fuse, + the fused MinkUNet-18 workload) or 'none'")
    ap.add_argument("--device-input", action="store_true",
                    help="the `device_input` record for THIS run's dtype: every timed step starts from the raw (120000, 4) scans in HBM "
                         "(round / shift / sparse_quantize / collate of all frames on the device); the default fp32 run carries the bf16 one")
    ap.add_argument("--no-device-input-line", action="store_true", help="skip the secondary `device_input` record")
    ap.add_argument("--verbose-json", action="store_true", help="print the long records (notes, clocks) instead of the compact line")
    args = ap.parse_args()
    if args.cpu_baseline_worker:
        _cpu_baseline_worker()
        return
    if args.cpu_pytorch_worker:
        _cpu_pytorch_worker()
        return
    self_launch(args)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s) (WORLD_SIZE): refusing to print a line for "
                         "another job size" % (args.gpus, world))
    if os.environ.get("PCS_BENCH_LAUNCH_CHECK") == "1":  # test rig (no GPU needed): the process group only
        dist.init_process_group(backend="gloo")
        t = torch.ones(1)
        dist.all_reduce(t)
        if rank == 0:
            print(json.dumps({"n_gpus": dist.get_world_size(), "ranks_seen": int(t.item())}), flush=True)
        dist.destroy_process_group()
        return
    # PCS_BENCH_FORCE_DIST=1 (test rig): the distributed code path (process group, DDP, sync BatchNorm, barrier,
    # max-over-ranks) even for one rank, so that it runs over RCCL on a single-GPU box
    distributed = world > 1 or os.environ.get("PCS_BENCH_FORCE_DIST") == "1"
    # PCS_BENCH_ONE_DEVICE=1 (test rigs with a single GPU): every rank uses cuda:0 and gloo carries the
    # collectives, to exercise the N > 1 code path; real runs use one GPU per rank and RCCL.
    one_dev = os.environ.get("PCS_BENCH_ONE_DEVICE") == "1"
    dev_index = 0 if one_dev else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_dev:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=dev)  # RCCL over xGMI

    # frames are sharded by rank: each rank owns frames_per_gpu whole frames (weak scaling)
    seeds = [rank * args.frames_per_gpu + i for i in range(args.frames_per_gpu)]
    batch = to_device(make_batch(seeds), dev)
    n_vox = batch["lidar"].C.shape[0]
    be = native.backend()
    from openpcseg_amd import functional as pcsF

    raw_dev = None

    def measure(amp, conv="fp32", wgrad="fp32", device_input=False):
        """One bench line: fresh model / optimizer (same seed), preheat, W warm-up steps, K timed steps.
        device_input: every step starts from the RAW scans in HBM (dataset transform + voxel dedup + collate on the device,
        workloads/synthetic.py::device_collate) instead of the pre-voxelised batch."""
        nonlocal raw_dev
        if device_input and raw_dev is None:
            from openpcseg_amd.workloads.synthetic import make_raw_batch
            raw_dev = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in make_raw_batch(seeds).items()}
        pcsF.set_conv_policy(conv)
        pcsF.set_wgrad_policy(wgrad)
        torch.manual_seed(0)
        model = MinkUNet(num_class=20, num_layer=MK34_LAYERS, cr=1.0, dist=distributed).to(dev).train()
        if distributed:
            model = torch.nn.parallel.DistributedDataParallel(model, device_ids=[dev_index])
        params = [p for p in model.parameters() if p.requires_grad]
        opt = torch.optim.SGD(params, lr=0.02 * args.frames_per_gpu * world / 8, momentum=0.9, weight_decay=1e-4,
                              nesterov=True)
        amp_dtype = {"bf16": torch.bfloat16, "fp16": torch.float16}.get(amp)
        scaler = torch.amp.GradScaler("cuda") if amp == "fp16" else None  # the reference scales fp16 losses (train.py)

        pipe = None
        if device_input and os.environ.get("PCS_BENCH_DI_PREFETCH", "0") == "1":
            # opt-in: collate of the next batch on a side stream during the step (workloads/synthetic.py::DeviceInputPrefetcher).
            # Measured [r6, profiles/round6_device_input_ab.txt]: 3.0 ms per step against 1.5 ms for the collate inside the step --
            # the side stream's sort kernels slow the step's own more than the host read they hide -- so the record times the
            # in-step form
            from openpcseg_amd.workloads.synthetic import DeviceInputPrefetcher
            pipe = DeviceInputPrefetcher(lambda i: raw_dev)

        def inputs():
            if device_input:
                if pipe is not None:
                    return pipe.next()
                from openpcseg_amd.workloads.synthetic import device_collate
                return device_collate(raw_dev)
            return fresh(batch)

        def step():
            loss = step_body()
            if device_input and pipe is not None:
                pipe.prefetch()
            return loss

        def step_body():
            opt.zero_grad(set_to_none=True)
            if amp is None:
                out = model(inputs())
                out["loss"].backward()
            else:
                with torch.autocast("cuda", dtype=amp_dtype):
                    out = model(inputs())
                if scaler is not None:
                    scaler.scale(out["loss"]).backward()
                    scaler.unscale_(opt)
                    torch.nn.utils.clip_grad_norm_(params, 10.0)
                    scaler.step(opt)
                    scaler.update()
                    return out["loss"]
                out["loss"].backward()
            torch.nn.utils.clip_grad_norm_(params, 10.0)
            opt.step()
            return out["loss"]

        with ConvMeter(be) as meter:
            preheat(step, distributed, dev)
            for _ in range(args.warmup):
                step()
            if distributed:
                dist.barrier()
            torch.cuda.synchronize()
            base_dt = dev_dt = None
            if device_input:
                # the same model from the pre-voxelised batch and from the raw scans in ALTERNATING blocks of K steps (three of each);
                # the record is the difference of the two medians. Round 5 timed one block of each back to back: at K = 5 a block's
                # time moves by +-1 ms per step from block to block (the driver's run read 7.0 ms where the builder's read 1.9-2.9)
                blocks = {False: [], True: []}
                for _ in range(3):
                    for flag in (False, True):
                        device_input = flag
                        step()
                        torch.cuda.synchronize()
                        t0 = time.perf_counter()
                        for _ in range(args.steps):
                            step()
                        torch.cuda.synchronize()
                        blocks[flag].append(time.perf_counter() - t0)
                base_dt, dev_dt = sorted(blocks[False])[1], sorted(blocks[True])[1]
                device_input = True
            meter.enabled = True
            with ClockSampler(dev_index) as clocks:
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    loss = step()
                torch.cuda.synchronize()
                if distributed:
                    dist.barrier()
                dt = time.perf_counter() - t0
            if dev_dt is not None:
                dt = dev_dt   # the median block (this last block served the conv meter and the clock sampler)
            meter.enabled = False
            roof = meter.summary(amp, split=(conv == "bf16x3"))
            clk = clocks.summary()
            if roof is not None:
                roof["clock"] = clk
                if clk.get("sclk_mhz_mean") and roof["bound"] == "mfma":
                    # the nominal peak is quoted at 2400 MHz; what the MFMA pipe could deliver at the clock this box held
                    roof["peak_at_measured_clock"] = round(PEAK_FP32_MFMA_TFLOPS * clk["sclk_mhz_mean"] / 2400.0, 1)
                    roof["frac_at_measured_clock"] = round(roof["achieved"] / roof["peak_at_measured_clock"], 4)
        if distributed:
            tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        comm = comm_profile(step, rank) if distributed and not one_dev else None
        pcsF.set_conv_policy("fp32")
        pcsF.set_wgrad_policy("fp32")
        frames = args.frames_per_gpu * world * args.steps
        return {"value": round(frames / dt, 3), "ms_per_step": round(dt / args.steps * 1e3, 2),
                "loss": round(float(loss.detach()), 4), "roofline": roof, "comm": comm,
                "base_ms_per_step": None if base_dt is None else round(base_dt / args.steps * 1e3, 2)}

    def workload(amp):
        return ("MinkUNet-34 cr1.0 train step (fwd + CE/Lovasz + bwd + SGD), SemanticKITTI-shape synthetic scans "
                "(120k pts, 0.05 m voxels), %s" % ("fp32" if amp is None else
                                                   "autocast %s (16-bit MFMA convs, fp32 accumulate / master weights / "
                                                   "wgrad / statistics)" % amp))

    amp = None if args.amp == "off" else args.amp
    # the reference trains under --amp (R:dist_train.sh:18): the default run carries the bf16 step as a secondary record. It is
    # measured FIRST [r5]: whatever a fresh process / box still ramps (see preheat) is not charged to the headline record
    second = measure("bf16") if amp is None and not args.no_amp_line else None
    # headline: fp32 storage and fp32 MFMA arithmetic throughout (convolutions, weight gradient) unless --wgrad says otherwise
    head = measure(amp, wgrad=args.wgrad if amp is None else "fp32")
    # third record: the fp32 step with BOTH fp32-grade split policies (forward / input-gradient convolutions and the weight gradient
    # of the wide layers with fp32 operands as three bf16 planes on the 16-bit MFMAs); opt-in arithmetic, never the headline value
    third = measure(None, conv="bf16x3", wgrad="bf16x3") if amp is None and not args.no_split_line else None
    # SURVEY section 8 f1 with its number: the same step (bf16: the fastest step, where input work shows most) starting from the RAW
    # scans resident in HBM -- dataset transform + voxel dedup + collate of the 12 frames on the device inside every timed step
    dev_in = None
    if args.device_input or (amp is None and world == 1 and not args.no_amp_line and not args.no_device_input_line):
        d_amp = amp if args.device_input else "bf16"
        base = head if d_amp == amp else second
        di = measure(d_amp, device_input=True)
        from openpcseg_amd.workloads.synthetic import device_collate
        for _ in range(3):
            device_collate(raw_dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            device_collate(raw_dev)
        torch.cuda.synchronize()
        dev_in = {"dtype": d_amp or "f32", "value": di["value"], "ms_per_step": di["ms_per_step"],
                  "input_ms_per_step": round(di["ms_per_step"] - di["base_ms_per_step"], 2),   # vs the same model from the pre-voxelised batch, back to back
                  "collate_ms": round((time.perf_counter() - t0) / 20 * 1e3, 2),              # the device pass alone, 12 frames
                  "loss": di["loss"]}
    models = None
    if world == 1 and args.models != "none" and (args.models != "auto" or amp is None):
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        try:
            import modelbench
            models = modelbench.run(args.models, dev, steps=min(args.steps, 10), warmup=min(args.warmup, 2))
        except Exception as e:  # e.g. the reference sources are not staged on this box: the record says so
            models = {"error": (type(e).__name__ + ": " + str(e))[:160]}
    if rank == 0:
        def roof_compact(r):
            if r is None:
                return None
            keep = ("bound", "achieved", "peak", "unit", "frac", "traffic", "launches", "avg_launch_us", "mfma_tflops")
            out = {k: r[k] for k in keep if k in r}
            out["kernel"] = r["kernel"].split(" (")[0]
            clk = (r.get("clock") or {}).get("sclk_mhz_mean")
            if clk:
                out["sclk_mhz"] = clk
            return out
        pick = (lambda r: r) if args.verbose_json else roof_compact
        res = {
            "metric": "LiDAR frames/sec training MinkUNet-34 SemanticKITTI",
            "value": head["value"], "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": head["ms_per_step"], "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": amp or "f32", "data": "synthetic",
            "config": {"workload": "MinkUNet-34 cr1.0 train step, %d x 120k-pt synthetic SemanticKITTI scans/GPU, %s"
                                   % (args.frames_per_gpu, "fp32" if amp is None else "autocast " + amp),
                       "global_batch": args.frames_per_gpu * world, "voxels_per_gpu_batch": n_vox, "parallelism": "dp%d" % world,
                       "wgrad": "fp32" if amp is not None else args.wgrad, "loss": head["loss"], "notes": "profiles/bench_notes.json"},
            "roofline": pick(head["roofline"]),
        }
        if head["comm"] is not None:
            res["comm"] = head["comm"]
        if second is not None:
            res["amp_bf16"] = {"value": second["value"], "ms_per_step": second["ms_per_step"], "dtype": "bf16",
                               "roofline": pick(second["roofline"])}
            if second["comm"] is not None:
                res["amp_bf16"]["comm"] = second["comm"]
        if third is not None:
            r3 = third["roofline"] or {}
            res["fp32_bf16x3"] = {"value": third["value"], "ms_per_step": third["ms_per_step"], "dtype": "f32",
                                  "conv_tflops": r3.get("achieved"), "conv_frac_of_bf16_peak_over_6": r3.get("frac")}
            if args.verbose_json:
                res["fp32_bf16x3"]["roofline"] = third["roofline"]
        if dev_in is not None:
            res["device_input"] = dev_in
        if models is not None:
            if not args.verbose_json and "error" not in models:
                # frames/s of one training step per model (ms per step and batch sizes: --verbose-json, bench_notes.json)
                short = {"fmt": "frames/s [ref f32,ref bf16,+fuse f32,+fuse bf16] (reference source; + openpcseg_amd.fuse)"}
                r1 = lambda v: None if v is None else round(v, 1)
                for k, v in models.items():
                    name, src = k.split("/")
                    if src == "fused":
                        short[k] = [r1(v.get("f32")), r1(v.get("bf16"))]
                        continue
                    row = short.setdefault(name, [None] * 4)
                    o = 0 if src == "ref" else 2
                    row[o], row[o + 1] = r1(v.get("f32")), r1(v.get("bf16"))
                models = short
            res["models"] = models
        if world == 1 and not args.no_cpu_baseline and amp is None:
            res["cpu_baseline"] = cpu_baseline(compact=not args.verbose_json)
        print(json.dumps(res, separators=(",", ":")) if args.verbose_json else fit_line(res), flush=True)
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
