"""N > 1 path on CPU: world_size-2 gloo. Frames are sharded by rank (whole frames per rank);
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


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    _patch()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    net = _make_net()
    ddp = torch.nn.parallel.DistributedDataParallel(net)
    _loss(ddp, _frame(rank)).backward()
    grads = [p.grad.clone() for p in net.parameters()]
    if rank == 0:
        q.put([g.numpy() for g in grads])
    dist.barrier()
    dist.destroy_process_group()


def test_ddp_gradients_match_concatenated_batch(oracle_backend):
    from openpcseg_amd.hostdata import sparse_collate
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    ddp_grads = q.get(timeout=300)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    # single process, both frames concatenated (batch index 0 / 1 keeps them apart in every map)
    from openpcseg_amd.sparse import SparseTensor
    frames = [_frame(0), _frame(1)]
    both = sparse_collate([SparseTensor(f.F, f.C[:, :3]) for f in frames])
    net = _make_net()
    _loss(net, both).backward()
    for g_ddp, p in zip(ddp_grads, net.parameters()):
        # DDP averages over ranks; the concatenated loss is the SUM of the per-frame losses
        assert np.allclose(2.0 * g_ddp, p.grad.numpy(), rtol=1e-4, atol=1e-5)

