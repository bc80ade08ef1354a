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
Prints ONE JSON line (rank 0) with the throughput, the roofline of the dominant kernel
(fused gather-GEMM conv, fp32 MFMA) measured live with HIP events on the launch stream, and
the reference's CPU backend timed on the host cores for a bounded sample (baseline only).
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
PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E spec peak (6.3 TB/s achievable)
POINTS_PER_FRAME = 120000


class ConvMeter:
    """Wraps the backend's fused conv launches (fp32: conv_gather_gemm, half: conv_gather_gemm_h): HIP events on the
    launch stream (torch's current stream IS the stream the C ABI launches on) + the algorithmic work per launch,
    2*P*Cin*Cout flop and e*(Nin*Cin + Nout*Cout) + 8P + e*K*Cin*Cout bytes."""

    def __init__(self, be):
        self.be = be
        self.orig = {"f32": be.conv_gather_gemm, "half": be.conv_gather_gemm_h}
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
        self.be.conv_gather_gemm, self.be.conv_gather_gemm_h = wrapped32, wrapped16
        return self

    def __exit__(self, *a):
        self.be.conv_gather_gemm, self.be.conv_gather_gemm_h = self.orig["f32"], self.orig["half"]

    def summary(self, amp):
        kind = "half" if amp else "f32"
        recs = [r for r in self.records if r[5] == kind]
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
        if amp:
            # 16-bit MFMA is 16x the fp32 rate: the fused conv is bound by bytes ("HBM-bound in bf16"),
            # priced on its ALGORITHMIC bytes (each feature row once, weights once, rulebook once)
            gbs = abytes / (ms * 1e-3) / 1e9
            return dict({"kernel": "conv_os5h_kernel (pcs_conv_gather_gemm_h: fwd + dgrad, %s)" % amp, "bound": "hbm",
                         "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4),
                         "traffic": None, "mfma_tflops": round(tflops, 1),
                         "traffic_note": "no PMC pass for the half kernels yet; the operand stream (gathered rows re-read "
                                         "per offset, weights per row-block group) comes out of L2"}, **common)
        # HBM bytes per launch come from separate rocprofv3 --pmc passes (tools/conv_traffic.sh; rocprofv3 cannot run
        # inside bench.py). The file names the kernel revision it was measured on; a stale file is refused.
        traffic, note = None, "no PMC traffic file for this kernel revision (run tools/conv_traffic.sh)"
        tfile = os.path.join(ROOT, "profiles", "round2_conv_traffic.json")
        rev = native.load_library().pcs_conv_kernel_revision().decode()
        if os.path.exists(tfile):
            tj = json.load(open(tfile))
            if tj.get("kernel_revision") == rev:
                traffic = tj.get("hbm_bytes_per_launch")
                note = ("HBM bytes/launch, rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE (separate passes, FETCH doubled per the "
                        "gfx950 note), profiles/round2_conv_traffic.json, kernel revision " + rev)
            else:
                note = "profiles/round2_conv_traffic.json was measured on kernel revision %s, this build is %s: refused" % (
                    tj.get("kernel_revision"), rev)
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
CPU_BASELINE_SIZES = (3000, 9000)  # rays of the two bounded samples (~10 s + ~30 s of CPU work)
CPU_BASELINE_TIMEOUT_S = 240  # hard cap; the default bench run must finish within minutes


def _cpu_baseline_worker():
    """Runs in a subprocess (OMP_NUM_THREADS fixed before any OpenMP runtime starts): the reference's own compiled CPU
    backend (oracle/_ref) under the same MinkUNet-34 cr1.0 training step (fwd + loss + bwd) on ONE frame subsampled to
    two sizes. The cost per ray is not constant (sparser scans have fewer pairs per voxel, fixed per-call overheads),
    so the full-frame time is extrapolated with the exponent fitted to the two samples, t = a * rays^b, instead of
    linearly from one -- and reconciled with the one-off full-frame measurement in
    profiles/round2_cpu_baseline_full_frame.json (226.5 s per frame on the build container's 8 cores)."""
    import math
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
        out["loss"].backward()
        dt = time.perf_counter() - t0
        model.zero_grad(set_to_none=True)
        return dt

    run(500)  # warm-up: pages the backend and the OpenMP runtime in
    n1, n2 = CPU_BASELINE_SIZES
    t1, t2 = run(n1), run(n2)
    b = math.log(t2 / t1) / math.log(n2 / n1)
    b_used = min(max(b, 0.85), 1.15)  # two samples on a shared host: keep the extrapolation near-linear
    t_full = t2 * (POINTS_PER_FRAME / n2) ** b_used
    cores = CPU_BASELINE_THREADS if kind == "reference" else 1
    print(json.dumps({"value": round(1.0 / t_full, 5), "unit": "frames/s", "cores": cores, "kind": kind,
                      "sample": "1 frame (seed 0) subsampled to %d and %d of 120000 rays, MinkUNet-34 cr1.0 fwd+bwd once "
                                "each after a warm-up (%.1f s, %.1f s; %d threads); full-frame time extrapolated as "
                                "t ~ rays^b with the fitted b = %.2f (used %.2f) -> %.0f s per frame; one-off full-frame "
                                "measurement on the build container: 226.5 s (profiles/round2_cpu_baseline_full_frame.json)"
                                % (n1, n2, t1, t2, CPU_BASELINE_THREADS, b, b_used, t_full)}), flush=True)


def cpu_baseline():
    import subprocess
    env = dict(os.environ, OMP_NUM_THREADS=str(CPU_BASELINE_THREADS), MKL_NUM_THREADS=str(CPU_BASELINE_THREADS),
               HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", "1"],
                           env=env, capture_output=True, text=True, timeout=CPU_BASELINE_TIMEOUT_S)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if line:
            return json.loads(line[-1])
        return {"value": None, "unit": "frames/s", "cores": CPU_BASELINE_THREADS, "kind": "reference",
                "sample": "cpu baseline worker failed: " + (r.stderr.strip().splitlines() or ["?"])[-1][:200]}
    except subprocess.TimeoutExpired:
        return {"value": None, "unit": "frames/s", "cores": CPU_BASELINE_THREADS, "kind": "reference",
                "sample": "cpu baseline exceeded the %d s cap on this host" % CPU_BASELINE_TIMEOUT_S}


def preheat(step, distributed, dev, window=10, max_windows=40):
    """Untimed clock warm-up before the W warm-up steps. Some MI355X boxes start a fresh process well below their
    sustained clocks and take tens of seconds of load to get there (measured: the same binary at 173 -> 152 -> 137 ms
    per step over its first minute on one box -- about 0.5 % per second --, flat at 137 ms from the first step on
    others). Windows of `window` steps are run until a window is no longer > 0.4 % faster than the one before (at
    least three windows, at most `max_windows`); with several ranks the decision is shared so that every rank runs the
    same number of steps.
    The K timed steps that follow are full, unmodified steps. PCS_BENCH_PREHEAT=0 skips this."""
    if os.environ.get("PCS_BENCH_PREHEAT", "1") == "0":
        return
    prev = None
    for wi in range(max_windows):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(window):
            step()
        torch.cuda.synchronize()
        cur = (time.perf_counter() - t0) / window
        if os.environ.get("PCS_BENCH_PREHEAT_LOG") == "1":
            print("preheat window: %.1f ms/step" % (cur * 1e3), file=sys.stderr, flush=True)
        go = 1 if (wi < 2 or cur < 0.996 * prev) else 0
        if distributed:
            flag = torch.tensor([go], device=dev, dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            go = int(flag.item())
        prev = cur
        if not go:
            break


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames-per-gpu", type=int, default=12)  # BATCH_SIZE_PER_GPU of minkunet_mk34_cr10.yaml
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--amp", choices=["off", "bf16", "fp16"], default="off",
                    help="mixed precision like the reference's --amp (second bench line; the headline metric is fp32)")
    ap.add_argument("--cpu-baseline-worker", type=int, default=0, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_baseline_worker:
        _cpu_baseline_worker()
        return

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
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

    torch.manual_seed(0)
    model = MinkUNet(num_class=20, num_layer=MK34_LAYERS, cr=1.0, dist=distributed).to(dev).train()
    if distributed:
        model = torch.nn.parallel.DistributedDataParallel(model, device_ids=[dev_index])
    params = [p for p in model.parameters() if p.requires_grad]
    opt = torch.optim.SGD(params, lr=0.02 * args.frames_per_gpu * world / 8, momentum=0.9, weight_decay=1e-4,
                          nesterov=True)

    # frames are sharded by rank: each rank owns frames_per_gpu whole frames (weak scaling)
    seeds = [rank * args.frames_per_gpu + i for i in range(args.frames_per_gpu)]
    batch = to_device(make_batch(seeds), dev)
    n_vox = batch["lidar"].C.shape[0]

    amp = None if args.amp == "off" else args.amp
    amp_dtype = {"bf16": torch.bfloat16, "fp16": torch.float16}.get(args.amp)
    scaler = torch.amp.GradScaler("cuda") if args.amp == "fp16" else None  # the reference scales fp16 losses (train.py)

    def step():
        opt.zero_grad(set_to_none=True)
        if amp is None:
            out = model(fresh(batch))
            out["loss"].backward()
        else:
            with torch.autocast("cuda", dtype=amp_dtype):
                out = model(fresh(batch))
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

    be = native.backend()
    with ConvMeter(be) as meter:
        preheat(step, distributed, dev)
        for _ in range(args.warmup):
            step()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()
        meter.enabled = True
        with ClockSampler(dev_index) as clocks:
            t0 = time.perf_counter()
            for _ in range(args.steps):
                loss = step()
            torch.cuda.synchronize()
            if distributed:
                dist.barrier()
            dt = time.perf_counter() - t0
        meter.enabled = False
        roof = meter.summary(amp)
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

    if rank == 0:
        frames = args.frames_per_gpu * world * args.steps
        res = {
            "metric": "LiDAR frames/sec training MinkUNet-34 SemanticKITTI",
            "value": round(frames / dt, 3), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": amp or "f32", "data": "synthetic",
            "config": {"workload": "MinkUNet-34 cr1.0 train step (fwd + CE/Lovasz + bwd + SGD), SemanticKITTI-shape "
                                   "synthetic scans (120k pts, 0.05 m voxels), %s"
                                   % ("fp32" if amp is None else "autocast %s (16-bit MFMA convs, fp32 accumulate / master "
                                                                 "weights / wgrad / statistics)" % amp),
                       "frames_per_gpu": args.frames_per_gpu, "global_batch": args.frames_per_gpu * world,
                       "voxels_per_gpu_batch": n_vox, "parallelism": "dp%d" % world, "loss": round(float(loss.detach()), 4)},
            "roofline": roof,
        }
        if world == 1 and not args.no_cpu_baseline and amp is None:
            res["cpu_baseline"] = cpu_baseline()
        print(json.dumps(res), flush=True)
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
