"""N > 1 path on CPU: gloo at world sizes 2, 4 and 8, unequal frames per rank. Frames are sharded by rank (whole frames per rank);
the only exchange is the gradient all-reduce done by DDP. Checked: DDP gradients (mean over ranks) equal the
single-process gradients of the concatenated 2-frame batch. The sparse ops run on the CPU oracle here
(monkeypatched backend, test only) -- what is under test is the host logic: per-rank maps never mix
frames, autograd wiring, DDP bucketing over our Conv3d parameters."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_net():
    from openpcseg_amd import modules as spnn
    torch.manual_seed(1)
    return torch.nn.Sequential(spnn.Conv3d(4, 8, 3), spnn.ReLU(True), spnn.Conv3d(8, 8, 2, stride=2), spnn.ReLU(True),
                               spnn.Conv3d(8, 6, 3), spnn.ReLU(True), spnn.Conv3d(6, 5, 2, stride=2, transposed=True))


def _frame(seed):
    from openpcseg_amd.workloads.synthetic import make_batch
    b = make_batch([seed], n_points=600, voxel_size=0.4)
    return b["lidar"]


def _patch():
    sys.path.insert(0, ROOT)
    from oracle.adapter import OracleBackend
    from openpcseg_amd import native
    native._BACKEND = OracleBackend()


def _loss(net, lidar):
    from openpcseg_amd.sparse import SparseTensor
    out = net(SparseTensor(lidar.F.clone(), lidar.C.clone()))
    return (out.F ** 2).sum()


def _frames_of(rank, world):
    """Unequal shards: odd ranks hold two frames, even ranks one (whole frames per rank)."""
    first = sum(1 + (r % 2) for r in range(rank))
    return list(range(first, first + 1 + (rank % 2)))


def _local_batch(seeds):
    from openpcseg_amd.hostdata import sparse_collate
    from openpcseg_amd.sparse import SparseTensor
    frames = [_frame(s) for s in seeds]
    return sparse_collate([SparseTensor(f.F, f.C[:, :3]) for f in frames])


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    _patch()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    net = _make_net()
    ddp = torch.nn.parallel.DistributedDataParallel(net)
    _loss(ddp, _local_batch(_frames_of(rank, world))).backward()
    grads = [p.grad.clone() for p in net.parameters()]
    if rank == 0:
        q.put([g.numpy() for g in grads])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_ddp_gradients_match_concatenated_batch(oracle_backend, world):
    """R:train.py:215-219 at world sizes 2, 4 and 8 with UNEQUAL frames per rank: DDP's mean over the ranks x world = the gradient
    of the summed loss of all frames in one process."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    ddp_grads = q.get(timeout=600)
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0
    # single process, all frames concatenated (the batch index keeps them apart in every map)
    both = _local_batch([s for r in range(world) for s in _frames_of(r, world)])
    net = _make_net()
    _loss(net, both).backward()
    for g_ddp, p in zip(ddp_grads, net.parameters()):
        # DDP averages over ranks; the concatenated loss is the SUM of the per-frame losses
        assert np.allclose(float(world) * g_ddp, p.grad.numpy(), rtol=2e-4, atol=2e-5)



def _bn_cuts(world):
    """Ragged row shards of the 300-row batch."""
    w = np.array([1 + (3 * r) % 5 for r in range(world)], dtype=np.float64)
    cuts = np.concatenate([[0], np.round(np.cumsum(w) / w.sum() * 300)]).astype(int)
    return cuts.tolist()


def _bn_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    _patch()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from openpcseg_amd.fused import FusedBatchNorm
    from openpcseg_amd.sparse import SparseTensor
    torch.manual_seed(5)
    full = torch.randn(300, 6) * 2 + 1
    res = torch.randn(300, 6)
    cuts = _bn_cuts(world)
    lo, hi = cuts[rank], cuts[rank + 1]                     # ragged shards
    x = full[lo:hi].clone().requires_grad_(True)
    bn = FusedBatchNorm(6, sync=True).train()
    y = bn(SparseTensor(x, torch.zeros(hi - lo, 4, dtype=torch.int32)), residual=res[lo:hi], relu=True).F
    (y * torch.arange(1, 7)).sum().backward()
    q.put((rank, y.detach().numpy(), x.grad.numpy(), bn.weight.grad.numpy(), bn.running_var.numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_sync_fused_batchnorm_matches_global_batch(oracle_backend, world):
    """SyncBN semantics of FusedBatchNorm(sync=True): `world` ragged shards == one BatchNorm1d over all rows."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bn_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=600) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0
    torch.manual_seed(5)
    full = (torch.randn(300, 6) * 2 + 1).requires_grad_(True)
    res = torch.randn(300, 6)
    ref = torch.nn.BatchNorm1d(6).train()
    y = torch.relu(ref(full) + res)
    (y * torch.arange(1, 7)).sum().backward()
    y_sh = np.concatenate([g[1] for g in got])
    gx_sh = np.concatenate([g[2] for g in got])
    assert np.allclose(y_sh, y.detach().numpy(), atol=1e-5)
    assert np.allclose(gx_sh, full.grad.numpy(), atol=1e-5)
    assert np.allclose(sum(g[3] for g in got), ref.weight.grad.numpy(), atol=1e-4)   # local sums add up (DDP averages)
    for g in got:
        assert np.allclose(g[4], ref.running_var.numpy(), rtol=1e-5)


@pytest.mark.gpu
def test_bench_two_ranks_on_one_device():
    """The N > 1 code path of bench.py on the HIP backend: two ranks launched exactly like the driver launches them
    (torch.distributed.run), both on cuda:0 with gloo carrying the collectives (PCS_BENCH_ONE_DEVICE rig): DDP over the
    sparse convolutions, FusedBatchNorm in sync mode (per-layer statistics all-reduce, cached global row counts),
    barrier + max-over-ranks timing, one JSON line from rank 0 with the whole-job frame count."""
    import json
    import subprocess
    env = dict(os.environ, PCS_BENCH_ONE_DEVICE="1", MASTER_ADDR="127.0.0.1", PCS_BENCH_PREHEAT_MIN_S="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--frames-per-gpu", "2", "--no-amp-line", "--no-split-line"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["scaling"] == "weak" and res["config"]["global_batch"] == 4
    assert res["value"] > 0 and np.isfinite(res["config"]["loss"]) and "cpu_baseline" not in res
    assert res["roofline"]["launches"] > 0


@pytest.mark.gpu
def test_bench_one_rank_over_rccl():
    """The distributed path of bench.py over RCCL (backend "nccl") on ONE MI355X: a one-rank job launched like the driver
    launches its ranks, forced through init_process_group("nccl"), DDP, FusedBatchNorm(sync=True) with its statistics
    all-reduce on the dedicated high-priority communicator (PCS_SYNC_WORLD1), barrier and the max-over-ranks all-reduce.
    The same frames without the process group must give the same loss."""
    import json
    import subprocess

    def run(env_extra, launcher, extra=()):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", PCS_BENCH_PREHEAT="0", **env_extra)  # same number of optimizer steps
        cmd = launcher + [os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--frames-per-gpu", "2",
                          "--no-cpu-baseline", "--models", "none"] + list(extra)
        out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=280)
        assert out.returncode == 0, out.stderr[-3000:]
        lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, out.stdout[-2000:]
        return json.loads(lines[0])

    rccl = run({"PCS_BENCH_FORCE_DIST": "1", "PCS_SYNC_WORLD1": "1"},
               [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                "--master-port", str(_free_port())])
    plain = run({}, [sys.executable], ["--no-amp-line", "--no-split-line"])
    assert rccl["n_gpus"] == 1 and rccl["value"] > 0 and rccl["roofline"]["launches"] > 0
    assert abs(rccl["config"]["loss"] - plain["config"]["loss"]) <= 2e-3 * abs(plain["config"]["loss"])
    # the default run carries the bf16 step as a secondary record (the reference trains under --amp)
    assert rccl["amp_bf16"]["value"] > 0 and rccl["amp_bf16"]["roofline"]["launches"] > 0 and "amp_bf16" not in plain
    assert rccl["fp32_bf16x3"]["value"] > 0 and "fp32_bf16x3" not in plain   # and the split-kernel fp32 step
    # device trace of one step of the distributed job: when RCCL launches kernels for the gradient buckets (a one-rank
    # communicator may be served by copies), the first of them must start before backward's last conv has finished --
    # i.e. DDP's bucketed all-reduce overlaps backward instead of trailing it
    comm = rccl.get("comm")
    assert comm is not None and "error" not in comm, comm
    print("one-rank RCCL comm record:", comm)
    if os.path.isdir(os.path.join(ROOT, "gpurun_out")):  # kept for profiles/: what the device trace of the RCCL job looked like
        with open(os.path.join(ROOT, "gpurun_out", "one_rank_rccl_bench.json"), "w") as f:
            json.dump({"comm": comm, "amp_bf16_comm": rccl["amp_bf16"].get("comm"), "value": rccl["value"],
                       "plain_value": plain["value"]}, f, indent=1)
    if comm.get("rccl_kernels", 0) > 0 and comm.get("first_rccl_kernel_before_last_conv_ends") is not None:
        assert comm["first_rccl_kernel_before_last_conv_ends"], comm


def _nccl_bn_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)  # RCCL
    from openpcseg_amd.fused import FusedBatchNorm
    from openpcseg_amd.sparse import SparseTensor
    torch.manual_seed(5)
    full = torch.randn(3000, 64) * 2 + 1
    res = torch.randn(3000, 64)
    lo, hi = (0, 1200) if rank == 0 else (1200, 3000)          # ragged shards
    x = full[lo:hi].to(dev).requires_grad_(True)
    bn = FusedBatchNorm(64, sync=True).to(dev).train()
    y = bn(SparseTensor(x, torch.zeros(hi - lo, 4, dtype=torch.int32, device=dev)), residual=res[lo:hi].to(dev), relu=True).F
    (y * torch.arange(1, 65, device=dev)).sum().backward()
    q.put((rank, y.detach().cpu().numpy(), x.grad.cpu().numpy(), bn.weight.grad.cpu().numpy(), bn.running_var.cpu().numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_sync_fused_batchnorm_over_rccl():
    """FusedBatchNorm(sync=True) on two MI355X over RCCL (backend "nccl"): statistics all-reduce on the dedicated
    process group, device-resident global count. Skipped on boxes with one visible GPU."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 visible GPUs (RCCL)")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_nccl_bn_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=300) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    torch.manual_seed(5)
    full = (torch.randn(3000, 64) * 2 + 1).requires_grad_(True)
    res = torch.randn(3000, 64)
    ref = torch.nn.BatchNorm1d(64).train()
    y = torch.relu(ref(full) + res)
    (y * torch.arange(1, 65)).sum().backward()
    assert np.allclose(np.concatenate([got[0][1], got[1][1]]), y.detach().numpy(), atol=1e-4)
    assert np.allclose(np.concatenate([got[0][2], got[1][2]]), full.grad.numpy(), atol=1e-4)
    assert np.allclose(got[0][3] + got[1][3], ref.weight.grad.numpy(), rtol=1e-4, atol=1e-3)
    assert np.allclose(got[0][4], ref.running_var.numpy(), rtol=1e-5)


@pytest.mark.gpu
def test_bench_two_ranks_over_rccl():
    """bench.py --gpus 2 exactly as the driver launches it, one GPU per rank, backend "nccl" (RCCL over xGMI): the first
    execution of the real multi-GPU path whenever the box shows two devices; skipped otherwise."""
    import json
    import subprocess
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 visible GPUs (RCCL)")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", PCS_BENCH_PREHEAT_MIN_S="0")
    env.pop("PCS_BENCH_ONE_DEVICE", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--frames-per-gpu", "2"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=560)
    assert out.returncode == 0, out.stderr[-2000:]
    res = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert res["n_gpus"] == 2 and res["config"]["global_batch"] == 4 and res["value"] > 0
    assert res["comm"]["rccl_kernels"] > 0 and res["comm"]["first_rccl_kernel_before_last_conv_ends"], res["comm"]
    assert res["amp_bf16"]["value"] > 0


def test_bench_gpus_flag_launches_the_ranks_itself():
    """`python bench.py --gpus 2` with no launcher around it starts two ranks (torch.distributed.run on 127.0.0.1, as
    R:dist_train.sh:17-19 does) and the line it prints says n_gpus = 2; under a launcher whose WORLD_SIZE disagrees with
    --gpus it refuses. PCS_BENCH_LAUNCH_CHECK=1: the process group only (gloo), no GPU needed."""
    import json
    import subprocess
    env = dict(os.environ, PCS_BENCH_LAUNCH_CHECK="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-1500:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    assert json.loads(line) == {"n_gpus": 2, "ranks_seen": 2}
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=dict(env, WORLD_SIZE="1", RANK="0"),
                         capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0 and "refusing" in (bad.stderr + bad.stdout)


# ---- block fusion of the reference's OWN blocks with the model file's SyncBatchNorm classes (openpcseg_amd/block_fusion.py) ---------
def _fused_block_inputs():
    from openpcseg_amd.workloads.synthetic import make_batch
    b = make_batch([21, 22], n_points=1500)
    torch.manual_seed(11)
    return b["lidar"].C, torch.randn(b["lidar"].C.shape[0], 16)


def _reference_residual_block(if_dist):
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import make_golden as mg
    import openpcseg_amd
    from seeded import seeded_state
    openpcseg_amd.install_reference_aliases()
    mod = mg.import_reference_model("pcseg.model.segmentor.voxel.minkunet.minkunet")
    blk = mod.ResidualBlock(16, 24, if_dist=if_dist)      # 16 -> 24 channels: the block with a downsample branch (1x1x1 conv + BN)
    seeded_state(blk)
    return blk.train()


def _fused_block_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    _patch()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import openpcseg_amd
    from openpcseg_amd.sparse import SparseTensor
    coords, feats = _fused_block_inputs()
    sel = coords[:, 3] == rank                           # whole frames per rank, like the DistributedSampler
    blk = _reference_residual_block(if_dist=True)
    counts = openpcseg_amd.fuse(blk)
    assert counts["residual"] == 1 and counts["conv_bn"] == 3, counts
    x = feats[sel].clone().requires_grad_(True)
    y = blk(SparseTensor(x, coords[sel].contiguous())).F
    (y * torch.arange(1, 25)).sum().backward()
    q.put((rank, y.detach().numpy(), x.grad.numpy(), blk.net[1].weight.grad.numpy(), blk.net[4].running_var.numpy(),
           int(blk.net[1].num_batches_tracked)))
    dist.barrier()
    dist.destroy_process_group()


def test_fused_reference_block_with_syncbatchnorm_matches_global_batch(oracle_backend):
    """R:pcseg/model/segmentor/voxel/minkunet/minkunet.py:83-129 with IF_DIST=True (the model file's SyncBatchNorm classes) after
    `openpcseg_amd.fuse`, two ranks with one frame each over gloo == the same block with plain BatchNorm over both frames in one
    process (the SyncBatchNorm contract): outputs, input gradients, parameter gradients (DDP would average
    them: per-rank sums add up to the global gradient), running statistics."""
    if not os.path.isdir("/root/reference") and not os.path.isdir(os.path.join(ROOT, "tests", "_refsrc")):
        pytest.skip("reference sources not present")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_fused_block_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=300) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    from openpcseg_amd.sparse import SparseTensor
    coords, feats = _fused_block_inputs()
    ref = _reference_residual_block(if_dist=False)        # plain BatchNorm1d, unfused, both frames at once
    x = feats.clone().requires_grad_(True)
    y = ref(SparseTensor(x, coords)).F
    (y * torch.arange(1, 25)).sum().backward()
    order = torch.cat([(coords[:, 3] == r).nonzero().squeeze(1) for r in range(2)])
    assert np.allclose(np.concatenate([got[0][1], got[1][1]]), y.detach().numpy()[order], atol=2e-5)
    assert np.allclose(np.concatenate([got[0][2], got[1][2]]), x.grad.numpy()[order], atol=2e-5)
    assert np.allclose(got[0][3] + got[1][3], ref.net[1].weight.grad.numpy(), rtol=1e-4, atol=1e-5)
    assert np.allclose(got[0][4], ref.net[4].running_var.numpy(), rtol=1e-5) and np.allclose(got[1][4], got[0][4])
    assert got[0][5] == got[1][5] == int(ref.net[1].num_batches_tracked) == 1


# ---- the reference's whole MinkUNet (IF_DIST=True) under fuse + DDP over two ranks vs one process on the concatenated batch ---------
_MK_DDP = dict(NAME="MinkUNet", IGNORE_LABEL=0, IN_FEATURE_DIM=4, BLOCK="ResBlock", NUM_LAYER=[1] * 8,
               PLANES=[32, 32, 64, 128, 256, 256, 128, 96, 96], cr=0.25, DROPOUT_P=0.0, LABEL_SMOOTHING=0.0)


def _ref_minkunet(if_dist):
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import make_golden as mg
    import openpcseg_amd
    from seeded import seeded_state
    openpcseg_amd.install_reference_aliases()
    mod = mg.import_reference_model("pcseg.model.segmentor.voxel.minkunet.minkunet")
    model = mod.MinkUNet(mg._AttrDict(dict(_MK_DDP, IF_DIST=if_dist)), 20)
    seeded_state(model)
    return model.train()


def _ddp_frames(world=2):
    from openpcseg_amd.workloads.synthetic import make_batch
    n = sum(len(_frames_of(r, world)) for r in range(world)) if world > 2 else 2
    return make_batch([31 + i for i in range(n)], n_points=1200)


def _rank_frames(rank, world):
    return [rank] if world <= 2 else _frames_of(rank, world)      # world 2: the round-5 case (one frame each); else unequal shards


def _fused_ddp_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if world > 2:
        torch.set_num_threads(1)
    _patch()
    torch.Tensor.cuda = lambda self, *a, **k: self        # the reference's forward calls .cuda() on the targets
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import openpcseg_amd
    from openpcseg_amd.sparse import SparseTensor
    b = _ddp_frames(world)
    mine = _rank_frames(rank, world)
    fid = b["lidar"].C[:, 3]
    sel = torch.zeros_like(fid, dtype=torch.bool)
    coords = b["lidar"].C.clone()
    for local, f in enumerate(mine):                      # whole frames per rank (DistributedSampler), batch index 0.. on each rank
        sel |= fid == f
        coords[fid == f, 3] = local
    coords = coords[sel]
    model = _ref_minkunet(if_dist=True)
    counts = openpcseg_amd.fuse(model)
    assert counts["residual"] == 8 and counts["forward"] == 1
    # DistributedDataParallel refuses host modules that hold SyncBatchNorm layers ("only work with GPU modules"), so its one data-path
    # action -- averaging the gradients over the ranks -- is done by hand here; the GPU form is bench.py's / the RCCL tests'
    ret = model({"lidar": SparseTensor(b["lidar"].F[sel].clone(), coords), "targets": SparseTensor(b["targets"].F[sel], coords), "offset": None})
    ret[0]["loss"].backward()
    for p in model.parameters():
        if p.grad is not None:
            dist.all_reduce(p.grad)
            p.grad /= world
    if rank == 0:
        q.put(({n: p.grad.numpy().copy() for n, p in model.named_parameters() if p.grad is not None},
               {n: t.numpy().copy() for n, t in model.named_buffers() if t.dtype.is_floating_point}))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_fused_reference_minkunet_under_ddp_matches_the_concatenated_batch(oracle_backend, monkeypatch, world):
    """What `R:train.py:215-219` builds -- gradient averaging around the model, SyncBatchNorm inside (IF_DIST=True) -- with the model
    fused by `openpcseg_amd.fuse`: two ranks with one frame each, and four ranks with 1 / 2 / 1 / 2 frames, over gloo. Per-frame mean losses averaged over the ranks = the gradient of
    0.5 (loss_0 + loss_1) with BatchNorm statistics over both frames: reproduced in ONE process by the unfused model with plain
    BatchNorm on the two-frame batch, backpropagating the two per-frame losses. Running statistics agree as well."""
    if not os.path.isdir("/root/reference") and not os.path.isdir(os.path.join(ROOT, "tests", "_refsrc")):
        pytest.skip("reference sources not present")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_fused_ddp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    grads, bufs = q.get(timeout=900)
    for p in procs:
        p.join(timeout=900)
        assert p.exitcode == 0
    from openpcseg_amd.sparse import SparseTensor
    monkeypatch.setattr(torch.Tensor, "cuda", lambda self, *a, **k: self)
    b = _ddp_frames(world)
    ref = _ref_minkunet(if_dist=False)
    crit = ref.criterion_losses
    seen = {}
    # the criterion is a per-frame mean on each rank: evaluate it per frame on the logits of the two-frame forward
    crit.register_forward_pre_hook(lambda m, a: seen.__setitem__("logits", a[0]))
    ref({"lidar": SparseTensor(b["lidar"].F.clone(), b["lidar"].C), "targets": SparseTensor(b["targets"].F, b["lidar"].C), "offset": None})
    logits, tgt, fid = seen["logits"], b["targets"].F.long(), b["lidar"].C[:, 3]
    loss = 0.0
    for r in range(world):
        mine = torch.zeros_like(fid, dtype=torch.bool)
        for f in _rank_frames(r, world):
            mine |= fid == f
        loss = loss + crit(logits[mine], tgt[mine]) / world
    loss.backward()
    G = float(np.median([float(p.grad.abs().max()) for p in ref.parameters() if p.grad is not None]))
    for n, p in ref.named_parameters():
        if p.grad is None:
            continue
        scale = max(float(p.grad.abs().max()), 1e-3 * G)
        assert np.abs(grads[n] - p.grad.numpy()).max() <= (2e-3 if world <= 2 else 4e-3) * scale, n   # four ranks: four summation orders
    for n, t in ref.named_buffers():
        if t.dtype.is_floating_point:
            assert np.allclose(bufs[n], t.numpy(), rtol=1e-4, atol=1e-6), n
