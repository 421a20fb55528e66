"""The N > 1 training path rehearsed on ONE GPU: two ranks share cuda:0 and exchange over gloo (NRHIP_DIST_BACKEND=gloo is
the same switch in bench.py).  It runs what a multi-GPU box runs -- process group, the gradient hooks of
GradientSynchronizer(overlap=True) on the real HIP autograd nodes (ProposalRoundFn / NffRenderTrainFn), HashGridAdam or
ShardedTableAdam on nrhip_adam_step -- and checks the data-parallel contract: after three steps on different ray shards the
replicas hold the parameters of ONE process trained on the mean loss.  Not a scaling measurement."""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _model():
    from neurad_studio_amd.models.neurad import NeuRADHotPath, NeuRADHotPathConfig

    c = NeuRADHotPathConfig(appearance_dim=16)
    c.field.sdf_beta = 3.0
    c.field.grid.static.log2_hashmap_size = 14
    c.sampling.proposal_field_1.grid.static.log2_hashmap_size = 13
    c.sampling.proposal_field_2.grid.static.log2_hashmap_size = 13
    torch.manual_seed(0)
    m = NeuRADHotPath(c, static_scale=100.0, num_sensors=3, duration=4.0).cuda().train()
    with torch.no_grad():
        m.field.hashgrid.static_grid.hash_table.mul_(1000.0)
        for p in m.proposal_fields:
            p.hashgrid.static_grid.hash_table.mul_(2000.0)
    m.sampler.eval()  # no jitter: both worlds walk the same samples
    return m


def _shard(rank, step, n=256):
    from neurad_studio_amd.cameras.rays import RayBundle

    g = torch.Generator().manual_seed(1000 * step + rank)
    o = torch.randn(n, 3, generator=g) * 5.0
    d = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1)
    return RayBundle(origins=o.cuda(), directions=d.cuda(), pixel_area=torch.full((n, 1), 2.7e-7, device="cuda"),
                     nears=torch.zeros(n, 1, device="cuda"), fars=None, times=(4 * torch.rand(n, 1, generator=g)).cuda(),
                     metadata={"sensor_idxs": torch.randint(0, 3, (n, 1), generator=g).cuda()})


def _loss(m, rb):
    from neurad_studio_amd.model_components.losses import distortion_loss, zipnerf_interlevel_loss

    out = m.get_nff_outputs(rb)
    return (out["features"].square().mean() + 1e-3 * out["depth"].mean()
            + 0.01 * zipnerf_interlevel_loss(out["weights_list"], out["ray_samples_list"])
            + 0.02 * distortion_loss(out["weights_list"], out["ray_samples_list"]))


def _optimizers(m, sharded):
    from neurad_studio_amd.optim import HashGridAdam
    from neurad_studio_amd.parallel.sharded_adam import ShardedTableAdam

    params = [p for p in m.parameters() if p.requires_grad]
    tables = [p for p in params if p.numel() >= 1 << 14]
    small = [p for p in params if p.numel() < 1 << 14]
    # eps = 1e-3, not the trainer's 1e-15: Adam divides by sqrt(v) + eps, so with a tiny eps an entry whose gradient is
    # rounding noise (1e-12 from a sample of weight ~0) still moves by +-lr, and the sign of that noise differs from run to
    # run (the MLP weight gradients are summed with atomics).  The test is about the exchange, not about that chaos.
    topt = ShardedTableAdam(tables, lr=1e-2, eps=1e-3, usage="static") if sharded else HashGridAdam(tables, lr=1e-2, eps=1e-3)
    return params, tables, topt, torch.optim.Adam(small, lr=1e-2, eps=1e-3)


def _worker(rank, world, port, mode, ret):
    import torch.distributed as dist

    from neurad_studio_amd.parallel.data_parallel import GradientSynchronizer

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m = _model()
    sharded = mode == "sharded"
    params, tables, topt, sopt = _optimizers(m, sharded)
    levels = None
    if mode == "level-sparse":  # coarse levels of the hash tables as (row, values) lists
        grids = [m.field.hashgrid.static_grid] + [p.hashgrid.static_grid for p in m.proposal_fields]
        levels = {g.hash_table: g.num_levels for g in grids}
    sync = GradientSynchronizer(params, average=True, large_threshold_bytes=1 << 14, usage="static", overlap=True,
                                skip=tables if sharded else (), level_tables=levels)
    lists = []
    overlapped = []
    level_runs = []
    for step in range(3):
        for o in (topt, sopt):
            o.zero_grad(set_to_none=True)
        _loss(m, _shard(rank, step)).backward()
        sync.sync()
        overlapped.append(sync.overlapped_last_step)
        lists.append(sum(len(v) for v in sync.last_list_levels.values()))
        level_runs.append(sync.overlapped_level_runs_last_step)
        topt.step(), sopt.step()
    torch.cuda.synchronize()
    if rank == 0:
        ret["overlapped"] = overlapped
        ret["lists"] = lists
        ret["level_runs"] = level_runs
        ret["params"] = {n: p.detach().cpu() for n, p in m.named_parameters()}
    ret[f"done{rank}"] = True
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["allreduce", "sharded", "level-sparse"],
                         ids=["allreduce+HashGridAdam", "ShardedTableAdam", "level-sparse exchange"])
def test_two_ranks_on_one_gpu_end_with_the_parameters_of_one_process_on_the_mean_loss(mode):
    import torch.multiprocessing as mp

    world = 2
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(world, _free_port(), mode, ret), nprocs=world, join=True)
        assert ret.get("done0") and ret.get("done1")
        got, overlapped, lists = dict(ret["params"]), list(ret["overlapped"]), list(ret["lists"])
        level_runs = list(ret["level_runs"])
    # one process, the mean loss of both shards
    m = _model()
    params, tables, topt, sopt = _optimizers(m, False)
    for step in range(3):
        for o in (topt, sopt):
            o.zero_grad(set_to_none=True)
        (sum(_loss(m, _shard(r, step)) for r in range(world)) / world).backward()
        topt.step(), sopt.step()
    worst = 0.0
    for n, p in m.named_parameters():
        a, b = got[n].double(), p.detach().cpu().double()
        err = float((a - b).norm() / (b.norm() + 1e-30))
        worst = max(worst, err)
        assert err < 2e-5, (n, err)
    # step 0 agrees on the usage set; afterwards the large gradients that every rank holds are exchanged from their hooks
    # (the field table and the one proposal table that trains; none when the sharded optimizer owns the tables)
    assert overlapped[0] == 0 and overlapped[1] == overlapped[2] == (2 if mode == "allreduce" else 0), overlapped
    if mode == "level-sparse":  # 256 rays x 32 samples into 2^14-row levels: the coarse levels of both tables go as lists
        assert min(lists) >= 2, lists
        # ... through csrc/grad_rows.hip (GPU gradients), and the levels that went densely in the previous step start their
        # reduce-scatter from the table's hook (20 levels in the three tables; the never-evaluated proposal table has none)
        assert level_runs[0] == 0 and (lists[0] >= 14 or min(level_runs[1:]) >= 1), (lists, level_runs)


def test_bench_gpus_2_as_a_plain_command_launches_its_own_ranks():
    """`python bench.py --gpus 2 ...` with NO launcher around it (WORLD_SIZE unset): bench.py spawns its two ranks itself
    (scripts/train.py:167-230 is the reference's own self-launch) and rank 0 prints the one JSON line.  Over gloo on this
    box's one GPU -- the labelled rehearsal, not a scaling number."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE", "GROUP_RANK")}
    env["NRHIP_DIST_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline", "--no-train"], env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and len(out["dist"]["devices"]) == 2 and out["dist"]["backend"] == "gloo"
    assert "rehearsal" in out and out["value"] > 0


def test_the_method_under_its_trainer_with_two_ranks():
    """`ns-train neurad-hip` at world_size 2, rehearsed on one GPU over gloo: `bench.py --gpus 2 --config c3 --via-plugin-only`
    steps NeuRADHipModel under HipTrainer (deferred scheduler step) + TableGradScaler on both ranks, with
    GradientSynchronizer(auto_sync=True) exchanging every gradient at the end of backward -- BEFORE the scaler's inf check, so
    that both ranks skip or step together -- and the reference's own Trainer.train_iteration over the same objects after it.
    Needs the reference (oracle/_ref).  Not a scaling number."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "oracle"))
    import ref_import

    if not ref_import.reference_available():
        pytest.skip("no reference (oracle/_ref ships with the lease)")
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE", "GROUP_RANK")}
    env["NRHIP_DIST_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--config", "c3", "--via-plugin-only",
                        "--steps", "3", "--warmup", "1", "--no-cpu-baseline"], env=env, cwd=root, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    via = out["train_via_plugin"]
    assert out["n_gpus"] == 2 and out["dist"]["backend"] == "gloo" and "error" not in via
    assert via["trainer"] == "HipTrainer" and via["grad_scaler"] == "TableGradScaler"
    assert via["optimizers"]["hashgrids"] == "HashGridAdam"
    assert via["grad_exchange_bytes_per_rank"] > 5e8  # the table gradients went through the exchange (0.56 GB per rank)
    assert via["grad_scaler_scale"] == 65536.0 and via["ms_per_iter"] > 0 and via["ms_per_iter_under_the_reference_trainer"] > 0
