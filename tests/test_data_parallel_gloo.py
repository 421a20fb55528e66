"""world_size-2 gloo tests (CPU) of the N>1 path: ray sharding and the flat-buffer gradient exchange
(SURVEY §8e).  The collective logic is device-agnostic; on the GPU box the same code runs over RCCL."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from neurad_studio_amd.parallel.data_parallel import GradientSynchronizer, shard_range


def test_shard_range_covers_everything_and_keeps_patches_whole():
    for n, world, gran in [(65536, 8, 1024), (57344, 8, 1), (40960 + 16384, 3, 1024), (10, 4, 1), (5, 8, 1)]:
        seen = []
        for r in range(world):
            s, e = shard_range(n, r, world, gran)
            assert s % gran == 0 or s == n
            seen += list(range(s, e))
        assert seen == list(range(n))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    # same replicated parameters on every rank; rank-dependent gradients; one parameter unused everywhere,
    # one used on rank 1 only (DDP find_unused_parameters semantics)
    table = torch.nn.Parameter(torch.zeros(4096, 4))       # "large": goes through reduce-scatter + all-gather
    w = torch.nn.Parameter(torch.zeros(33, 32))
    b = torch.nn.Parameter(torch.zeros(33))
    unused = torch.nn.Parameter(torch.zeros(7))
    partial = torch.nn.Parameter(torch.zeros(5))
    g = torch.Generator().manual_seed(100 + rank)
    table.grad = torch.randn(4096, 4, generator=g)
    w.grad = torch.randn(33, 32, generator=g)
    b.grad = torch.randn(33, generator=g)
    if rank == 1:
        partial.grad = torch.ones(5)
    sync = GradientSynchronizer([table, w, b, unused, partial], average=True, large_threshold_bytes=1 << 14)
    nbytes = sync.sync()
    # reference: plain all_reduce of the same tensors
    exp = []
    for r in range(world):
        gg = torch.Generator().manual_seed(100 + r)
        exp.append((torch.randn(4096, 4, generator=gg), torch.randn(33, 32, generator=gg), torch.randn(33, generator=gg)))
    ok = all(torch.allclose(p.grad, sum(e[i] for e in exp) / world, atol=1e-6) for i, p in enumerate((table, w, b)))
    ok = ok and unused.grad is None and torch.allclose(partial.grad, torch.full((5,), 1.0 / world))
    ok = ok and nbytes == (4096 * 4 + 33 * 32 + 33 + 5) * 4
    ret[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_synchronizer_world2_gloo():
    world = 2
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
        assert all(ret[r] for r in range(world)), dict(ret)
