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


class _TinyField(torch.nn.Module):
    """CPU stand-in with the hot path's parameter structure: two 'hash tables' (one of them never evaluated, like
    proposal_fields[0], models/neurad.py:248) + a small MLP."""

    def __init__(self):
        super().__init__()
        g = torch.Generator().manual_seed(1)
        self.table = torch.nn.Parameter(torch.randn(4096, 4, generator=g) * 0.1)
        self.prop_table = torch.nn.Parameter(torch.randn(2048, 4, generator=g) * 0.1)
        self.unused_table = torch.nn.Parameter(torch.randn(2048, 4, generator=g) * 0.1)
        self.mlp = torch.nn.Sequential(torch.nn.Linear(4, 8), torch.nn.ReLU(), torch.nn.Linear(8, 1))
        with torch.no_grad():
            for p in self.mlp.parameters():
                p.copy_(torch.randn(p.shape, generator=g) * 0.3)

    def forward(self, idx):
        prop = self.prop_table[idx % 2048].sum(-1).square().mean()  # an independent branch, like the interlevel loss
        return self.mlp(self.table[idx]).square().mean() + 0.1 * prop


def _worker_model(rank, world, port, ret, overlap):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m = _TinyField()
    opt = torch.optim.Adam(m.parameters(), lr=1e-2, eps=1e-15)
    sync = GradientSynchronizer(m.parameters(), average=True, large_threshold_bytes=1 << 14, usage="static", overlap=overlap)
    ref = _TinyField()  # single-process reference: the averaged loss of both shards
    ref_opt = torch.optim.Adam(ref.parameters(), lr=1e-2, eps=1e-15)
    overlapped = []
    for step in range(3):
        batches = [torch.randint(0, 4096, (512,), generator=torch.Generator().manual_seed(10 * step + r)) for r in range(world)]
        opt.zero_grad(set_to_none=True)
        m(batches[rank]).backward()
        sync.sync()
        overlapped.append(sync.overlapped_last_step)
        opt.step()
        ref_opt.zero_grad(set_to_none=True)
        (sum(ref(b) for b in batches) / world).backward()
        ref_opt.step()
    ok = all(torch.allclose(a, b, atol=1e-6) for a, b in zip(m.parameters(), ref.parameters()))
    ok = ok and m.unused_table.grad is None
    # step 0 agrees on the usage set without overlap; afterwards both large tables are exchanged from their hooks
    ok = ok and overlapped == ([0, 2, 2] if overlap else [0, 0, 0])
    # static usage: a parameter that suddenly gets a gradient is an error, not a silent divergence
    m.unused_table.grad = torch.zeros_like(m.unused_table)
    try:
        sync.sync()
        ok = False
    except RuntimeError:
        pass
    ret[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("overlap", [False, True])
def test_model_replicas_stay_identical_world2_gloo(overlap):
    """two model replicas, each on its own ray shard, one exchange per step (static usage set, optionally with the large
    tables' reduce-scatter launched from autograd hooks during the backward): parameters after 3 Adam steps equal a
    single process trained on the mean loss of both shards"""
    world = 2
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker_model, args=(world, _free_port(), ret, overlap), nprocs=world, join=True)
        assert all(ret[r] for r in range(world)), dict(ret)


def _torch_adam_update(param, grad, exp_avg, exp_avg_sq, step, lr, beta1, beta2, eps, weight_decay, grad_scale):
    """torch.optim.Adam's single-tensor arithmetic on a slice (stands in for nrhip_adam_step, which has no CPU path)"""
    g = grad * grad_scale
    if weight_decay:
        param.mul_(1 - lr * weight_decay)
    exp_avg.lerp_(g, 1 - beta1)
    exp_avg_sq.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1, bc2 = 1 - beta1 ** step, 1 - beta2 ** step
    denom = (exp_avg_sq.sqrt() / (bc2 ** 0.5)).add_(eps)
    param.addcdiv_(exp_avg, denom, value=-lr / bc1)


def _sharded_adam_worker(rank, world, port, ret):
    from neurad_studio_amd.parallel.sharded_adam import ShardedTableAdam

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    tables = [torch.nn.Parameter(torch.randn(1024, 4) * 0.1), torch.nn.Parameter(torch.randn(512, 2) * 0.1)]
    ref_tables = [torch.nn.Parameter(t.detach().clone()) for t in tables]
    ref_opt = torch.optim.Adam(ref_tables, lr=1e-2, eps=1e-15)  # one process on the MEAN gradient
    opt = ShardedTableAdam(tables, lr=1e-2, eps=1e-15, update_fn=_torch_adam_update)
    nbytes = 0
    for it in range(3):
        grads = []
        for r in range(world):
            g = torch.Generator().manual_seed(1000 * it + r)
            grads.append([torch.randn(t.shape, generator=g) for t in tables])
        for t, gr in zip(tables, grads[rank]):
            t.grad = gr.clone()
        if it == 1:  # rank 1 has no gradient for table 1 this step: it contributes zeros
            grads[1][1] = torch.zeros_like(grads[1][1])
            if rank == 1:
                tables[1].grad = None
        nbytes = opt.step()
        for i, t in enumerate(ref_tables):
            t.grad = sum(g[i] for g in grads) / world
        ref_opt.step()
    ok = all(torch.allclose(a, b, atol=1e-6, rtol=1e-5) for a, b in zip(tables, ref_tables))
    ok = ok and nbytes == sum(2 * t.numel() * 4 * (world - 1) // world for t in tables)
    sd = opt.state_dict()  # full-size moments in torch.optim.Adam's layout
    rsd = ref_opt.state_dict()
    for i in range(2):
        ok = ok and sd["state"][i]["exp_avg"].shape == tables[i].shape
        ok = ok and torch.allclose(sd["state"][i]["exp_avg"], rsd["state"][i]["exp_avg"], atol=1e-7)
        ok = ok and torch.allclose(sd["state"][i]["exp_avg_sq"], rsd["state"][i]["exp_avg_sq"], atol=1e-9)
        ok = ok and float(sd["state"][i]["step"]) == float(rsd["state"][i]["step"]) == 3.0
    # a fresh optimizer resumes from the gathered state (the reference's checkpoint layout) and keeps in step
    opt2 = ShardedTableAdam(tables, lr=1.0, update_fn=_torch_adam_update)
    opt2.load_state_dict(sd)
    for t, rt in zip(tables, ref_tables):
        g = torch.Generator().manual_seed(77)
        t.grad = torch.randn(t.shape, generator=g)
        rt.grad = t.grad.clone()  # identical on both ranks: the mean is the gradient itself
    opt2.step()
    ref_opt.step()
    ok = ok and all(torch.allclose(a, b, atol=1e-6, rtol=1e-5) for a, b in zip(tables, ref_tables))
    ret[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_table_adam_world2_matches_one_process_on_the_mean_gradient():
    world = 2
    ret = mp.get_context("spawn").Manager().dict()
    mp.spawn(_sharded_adam_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert all(ret[r] for r in range(world)), dict(ret)


def _uneven_overlap_worker(rank, world, port, ret):
    """overlap=True with a large table only ONE rank has a local gradient for (an actor grid no ray of the other rank
    hit): its reduce-scatter must not start from that rank's hook -- the other rank would issue it later, from sync(),
    and the two ranks' collective sequences would differ"""
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    a = torch.nn.Parameter(torch.randn(2048, 4))
    b = torch.nn.Parameter(torch.randn(2048, 4))  # same size as `a`: a mix-up would not even raise
    sync = GradientSynchronizer([a, b], average=True, large_threshold_bytes=1 << 14, usage="static", overlap=True)
    ok = True
    for step in range(3):
        ga = torch.full((2048, 4), float(rank + 1))
        gb = torch.full((2048, 4), 10.0)
        a.grad = b.grad = None
        loss = (a * ga).sum() + ((b * gb).sum() if rank == 0 else 0.0)  # b: a local gradient on rank 0 only
        loss.backward()
        sync.sync()
        ok = ok and torch.allclose(a.grad, torch.full((2048, 4), 1.5)) and torch.allclose(b.grad, torch.full((2048, 4), 5.0))
        ok = ok and sync.overlapped_last_step == (0 if step == 0 else 1)  # only `a` is exchanged from its hook
    # a second backward before sync() is an error, not a double reduction
    (a * 1.0).sum().backward()
    try:
        (a * 1.0).sum().backward()
        ok = False
    except RuntimeError:
        pass
    ret[rank] = bool(ok)
    sync._inflight.pop(0)[0].wait()
    dist.barrier()
    dist.destroy_process_group()


def test_overlap_hook_only_for_gradients_every_rank_holds_world2_gloo():
    world = 2
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_uneven_overlap_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
        assert all(ret[r] for r in range(world)), dict(ret)


def _sharded_adam_skip_worker(rank, world, port, ret):
    """a table without a gradient on ANY rank is skipped (parameters, moments and ITS step count untouched) -- what
    torch.optim.Adam does on grad is None; tables keep their own step counts through state_dict / load_state_dict"""
    from neurad_studio_amd.parallel.sharded_adam import ShardedTableAdam

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    tables = [torch.nn.Parameter(torch.randn(256, 4) * 0.1), torch.nn.Parameter(torch.randn(128, 4) * 0.1)]
    ref_tables = [torch.nn.Parameter(t.detach().clone()) for t in tables]
    ref_opt = torch.optim.Adam(ref_tables, lr=1e-2, eps=1e-15)
    opt = ShardedTableAdam(tables, lr=1e-2, eps=1e-15, update_fn=_torch_adam_update)
    for it in range(4):
        for i, (t, rt) in enumerate(zip(tables, ref_tables)):
            if i == 1 and it in (1, 2):  # the second table: no gradient anywhere in steps 1 and 2
                t.grad = rt.grad = None
                continue
            g = torch.randn(t.shape, generator=torch.Generator().manual_seed(10 * it + i))
            t.grad, rt.grad = g.clone(), g.clone()
        opt.step()
        ref_opt.step()
    ok = all(torch.allclose(a, b, atol=1e-6, rtol=1e-5) for a, b in zip(tables, ref_tables))
    sd, rsd = opt.state_dict(), ref_opt.state_dict()
    ok = ok and float(sd["state"][0]["step"]) == float(rsd["state"][0]["step"]) == 4.0
    ok = ok and float(sd["state"][1]["step"]) == float(rsd["state"][1]["step"]) == 2.0
    ok = ok and torch.allclose(sd["state"][1]["exp_avg"], rsd["state"][1]["exp_avg"], atol=1e-7)
    opt2 = ShardedTableAdam(tables, lr=1.0, update_fn=_torch_adam_update)
    opt2.load_state_dict(sd)
    ok = ok and [st["step"] for st in opt2.state] == [4, 2]
    ret[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_table_adam_skips_tables_without_any_gradient_world2_gloo():
    world = 2
    ret = mp.get_context("spawn").Manager().dict()
    mp.spawn(_sharded_adam_skip_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert all(ret[r] for r in range(world)), dict(ret)


def _worker_levels(rank, world, port, ret):
    """a 4-level table: level 0 gets 3 rows per rank, level 1 ~5 %, level 2 ~40 %, level 3 every row; a second table whose
    levels are all sparse; rank 1 has an all-zero level (empty list, padding only)"""
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    L, T, F = 4, 1024, 4
    g = torch.Generator().manual_seed(77 + rank)

    def grads():
        a = torch.zeros(L, T, F)
        for lvl, frac in enumerate((3 / T, 0.05, 0.4, 1.0)):
            rows = torch.randperm(T, generator=g)[:max(int(frac * T), 1)]
            a[lvl, rows] = torch.randn(len(rows), F, generator=g)
        a[0, 0] = 1.0 + rank  # row 0 in both lists: the padding's clamp target must survive
        b = torch.zeros(2, 512, 2)
        if rank == 0:
            b[0, torch.randperm(512, generator=g)[:20]] = torch.randn(20, 2, generator=g)
        b[1, torch.randperm(512, generator=g)[:7]] = torch.randn(7, 2, generator=g)
        return a.reshape(L * T, F), b.reshape(2 * 512, 2)

    ga, gb = grads()
    out = {}
    for mode in ("dense", "lists"):
        ta, tb = torch.nn.Parameter(torch.zeros(L * T, F)), torch.nn.Parameter(torch.zeros(1024, 2))
        small = torch.nn.Parameter(torch.zeros(5))
        ta.grad, tb.grad, small.grad = ga.clone(), gb.clone(), torch.full((5,), float(rank))
        sync = GradientSynchronizer([ta, tb, small], average=True, large_threshold_bytes=1 << 10,
                                    level_tables={ta: L, tb: 2} if mode == "lists" else None)
        sync.sync()
        out[mode] = (ta.grad.clone(), tb.grad.clone(), small.grad.clone(), sync.last_wire_bytes, dict(sync.last_list_levels))
    # the truth: mean of both ranks' gradients
    both = [torch.empty_like(ga) for _ in range(world)]
    dist.all_gather(both, ga)
    ok = torch.equal(out["dense"][0], (both[0] + both[1]) / 2)
    ok = ok and all(torch.equal(out["dense"][k], out["lists"][k]) for k in range(3))  # world 2: a + b in either order
    ok = ok and out["lists"][4] == {0: [0, 1, 2], 1: [0, 1]}  # level 2: 2 * 410 * 5 < 2 * 1024 * 4 still pays at N = 2
    ok = ok and out["lists"][3] < out["dense"][3]
    ret[rank] = (bool(ok), out["dense"][3], out["lists"][3])
    dist.barrier()
    dist.destroy_process_group()


def test_level_sparse_exchange_equals_the_dense_exchange_world2_gloo():
    world = 2
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker_levels, args=(world, _free_port(), ret), nprocs=world, join=True)
        assert all(ret[r][0] for r in range(world)), dict(ret)


def _worker_levels_overlap(rank, world, port, ret):
    """the level-sparse exchange with overlap=True: from the second step on the table's hook starts the count agreement and
    the dense levels' reduce-scatter; every step must still equal the plain dense exchange, on both ranks"""
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    L, T, F = 4, 1024, 4
    g = torch.Generator().manual_seed(177 + rank)

    def grads(step):
        a = torch.zeros(L, T, F)
        # level 2 turns dense in steps 3 and 4 (a spike) and collapses again from step 5 on
        for lvl, frac in enumerate((3 / T, 0.05, 0.9 if step in (3, 4) else (0.4 if step < 3 else 0.05), 1.0)):
            rows = torch.randperm(T, generator=g)[:max(int(frac * T), 1)]
            a[lvl, rows] = torch.randn(len(rows), F, generator=g)
        return a.reshape(L * T, F)

    ok, runs, lists = True, [], []
    ta = torch.nn.Parameter(torch.zeros(L * T, F))
    tb = torch.nn.Parameter(torch.zeros(L * T, F))  # the same gradients through the dense exchange
    small = torch.nn.Parameter(torch.zeros(5))
    sync_l = GradientSynchronizer([ta, small], average=True, large_threshold_bytes=1 << 10, usage="static", overlap=True,
                                  level_tables={ta: L})
    sync_d = GradientSynchronizer([tb], average=True, large_threshold_bytes=1 << 10, usage="static", overlap=True)
    for step in range(7):
        ga = grads(step)
        ta.grad = tb.grad = small.grad = None
        ((ta * ga).sum() + (tb * ga).sum() + small.sum() * (1.0 + rank)).backward()  # fires the post-accumulate hooks
        sync_l.sync(), sync_d.sync()
        ok = ok and torch.equal(ta.grad, tb.grad) and torch.equal(small.grad, torch.full((5,), 1.5))
        runs.append(sync_l.overlapped_level_runs_last_step)
        lists.append(list(sync_l.last_list_levels[0]))
    # step 0 agrees on the usage set (no hooks) and leaves its split behind: level 3 went densely, so from step 1 on its
    # reduce-scatter starts from the hook; in step 3 level 2 has turned dense too
    ok = ok and runs[0] == 0 and runs[1] == 1 and runs[2] == 1 and lists[0] == [0, 1, 2] and lists[2] == [0, 1, 2]
    ok = ok and lists[3] == [0, 1] and runs[3] == 1  # level 2 goes densely from sync() in step 3 (hook: only level 3)
    # step 4: the hook starts levels 2 + 3 as one dense run; step 5: level 2's count has collapsed, but its hook-started dense
    # exchange (decided from step 4's counts) is already on the wire -- it finishes densely and the NEXT step's dense set
    # is re-derived from step 5's counts alone; step 6: level 2 travels as a list again (dense -> list, not only list -> dense)
    ok = ok and lists[4] == [0, 1] and runs[4] == 1 and lists[5] == [0, 1] and runs[5] == 1
    ok = ok and lists[6] == [0, 1, 2] and runs[6] == 1
    ret[rank] = (bool(ok), runs, lists)
    dist.barrier()
    dist.destroy_process_group()


def test_level_sparse_exchange_with_hook_started_overlap_world2_gloo():
    world = 2
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker_levels_overlap, args=(world, _free_port(), ret), nprocs=world, join=True)
        assert all(ret[r][0] for r in range(world)), dict(ret)


def _wire16_worker(rank, world, port, ret):
    """wire_dtype=bf16: (a) GradientSynchronizer, direct and hook-started -- the exchanged gradient is bit-identical on both
    ranks, equals the fp32 mean of the ranks' bf16-ROUNDED gradients exactly, and is within bf16 rounding of the fp32
    exchange; (b) ShardedTableAdam -- parameters bit-identical across ranks and equal to one process stepping on that mean."""
    from neurad_studio_amd.parallel.sharded_adam import ShardedTableAdam

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ok = True
    shape = (2048, 4)

    def grad_of(r, it):
        return torch.randn(shape, generator=torch.Generator().manual_seed(500 + 10 * it + r))

    def gathered(t):
        parts = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(parts, t.contiguous())
        return parts

    for overlap in (False, True):
        ta, tb = torch.nn.Parameter(torch.zeros(shape)), torch.nn.Parameter(torch.zeros(shape))
        small = torch.nn.Parameter(torch.zeros(7))
        kw = dict(average=True, large_threshold_bytes=1 << 10, usage="static", overlap=overlap)
        s16 = GradientSynchronizer([ta, small], wire_dtype=torch.bfloat16, **kw)
        s32 = GradientSynchronizer([tb], **kw)
        for it in range(3):  # step 0 agrees on the usage set; with overlap the hooks start the exchange from step 1 on
            g = grad_of(rank, it)
            ta.grad = tb.grad = small.grad = None
            ((ta * g).sum() + (tb * g).sum() + small.sum() * (1.0 + rank)).backward()
            s16.sync(), s32.sync()
            want16 = sum(grad_of(r, it).bfloat16().float() for r in range(world)) / world
            ok = ok and torch.equal(ta.grad, want16)                       # fp32 sum of the once-rounded gradients
            ok = ok and all(torch.equal(p, ta.grad) for p in gathered(ta.grad))  # replicas bit-identical
            bound = sum(grad_of(r, it).abs() for r in range(world)) / world * 2.0**-8
            ok = ok and bool(((ta.grad - tb.grad).abs() <= bound + 1e-12).all())  # within bf16 rounding of the fp32 exchange
            ok = ok and torch.equal(small.grad, torch.full((7,), 1.5))      # small gradients keep fp32
            if overlap and it >= 1:
                ok = ok and s16.overlapped_last_step == 1
        ok = ok and s16.last_wire_bytes_by_param[0] == (world - 1) * ta.numel() * 6 // world
        ok = ok and s32.last_wire_bytes_by_param[0] == (world - 1) * tb.numel() * 8 // world
    # (b) the sharded optimizer on the 16-bit gradient leg
    torch.manual_seed(0)
    tables = [torch.nn.Parameter(torch.randn(1024, 4) * 0.1)]
    ref_tables = [torch.nn.Parameter(t.detach().clone()) for t in tables]
    ref_opt = torch.optim.Adam(ref_tables, lr=1e-2, eps=1e-15)
    opt = ShardedTableAdam(tables, lr=1e-2, eps=1e-15, update_fn=_torch_adam_update, wire_dtype=torch.bfloat16)
    for it in range(3):
        gs = [torch.randn(tables[0].shape, generator=torch.Generator().manual_seed(900 + 10 * it + r)) for r in range(world)]
        tables[0].grad = gs[rank].clone()
        nbytes = opt.step()
        ref_tables[0].grad = sum(g.bfloat16().float() for g in gs) / world
        ref_opt.step()
    ok = ok and torch.allclose(tables[0], ref_tables[0], atol=1e-6, rtol=1e-5)
    ok = ok and all(torch.equal(p, tables[0].data) for p in gathered(tables[0].data))
    ok = ok and nbytes == tables[0].numel() * 6 * (world - 1) // world
    ret[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


def test_bf16_wire_format_keeps_replicas_identical_world2_gloo():
    world = 2
    ret = mp.get_context("spawn").Manager().dict()
    mp.spawn(_wire16_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert all(ret[r] for r in range(world)), dict(ret)


def _auto_sync_worker(rank, world, port, ret):
    """the shape of the reference's iteration (engine/trainer.py:553-558): `grad_scaler.scale(loss).backward()` goes
    straight into `grad_scaler.step(optimizer)` -- nobody calls sync(); GradientSynchronizer(auto_sync=True) runs it at the
    end of the backward pass, BEFORE the scaler looks for infs.  Rank 1's batch overflows in step 1: both ranks must skip."""
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m = _TinyField()
    params = [p for p in m.parameters()]
    sync = GradientSynchronizer(params, average=True, large_threshold_bytes=1 << 14, usage="dynamic", auto_sync=True)
    opt = torch.optim.Adam(params, lr=1e-2)
    scaler = torch.amp.GradScaler("cpu", init_scale=8.0, growth_interval=1000)
    skipped = []
    for step in range(3):
        g = torch.Generator().manual_seed(10 * step + rank)
        idx = torch.randint(0, 4096, (256,), generator=g)
        x = m.table[idx] + m.prop_table[idx % 2048]
        loss = m.mlp(x).square().mean()
        if step == 1 and rank == 1:
            loss = loss * float("inf")
        opt.zero_grad(set_to_none=True)
        scaler.scale(loss).backward()
        before = m.table.detach().clone()
        scaler.step(opt)
        scaler.update()
        skipped.append(bool(torch.equal(before, m.table.detach())))
    ret[rank] = (skipped, [p.detach().clone() for p in params], scaler.get_scale(), sync.last_sync_bytes)
    dist.barrier()
    dist.destroy_process_group()


def test_auto_sync_reduces_before_the_gradscaler_looks_world2_gloo():
    world = 2
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_auto_sync_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
        (s0, p0, sc0, b0), (s1, p1, sc1, b1) = ret[0], ret[1]
    assert s0 == s1 == [False, True, False], (s0, s1)  # the overflow on rank 1 skipped the step on BOTH ranks
    assert sc0 == sc1 == 4.0 and b0 == b1 > 0
    for a, b in zip(p0, p1):
        assert torch.equal(a, b)  # replicas bit-identical
