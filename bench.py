#!/usr/bin/env python
"""bench.py -- BASELINE.json's metric.

    python bench.py --gpus 1 --steps K --warmup W [--config c1|c2|c3]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

--config c1 (default, the headline: BASELINE config[1], the configuration `metric` is quoted on).  One step = one pass of
  the hot path over a batch of synthetic rays resident in HBM: PowerSampler bins (S1) + ray ordering pass (one launch)
  -> fused hash-grid lookup + tiny MLPs (fp32 MFMA) + transmittance/alpha compositing (F1+C1+C2, nrhip_render_fwd_ex).
  4096 rays x 128 samples, HashEncoding(16 levels, T=2^19, F=2) + 64-wide MLPs, fp32 table.  Rays shard across ranks
  with no data-path collective (inference needs none) -> weak scaling.  The same JSON line carries
  `roofline` (dominant kernel, HIP-event timed inside the timed region), `train` (iters/s of the config-1 field),
  `train_full` (iters/s of the whole NeuRAD-default training step on BASELINE config[3]'s camera+lidar joint batch,
  40 960 + 16 384 rays per GPU, incl. losses, gradient exchange and Adam -- the train-iters/sec half of the metric) and,
  at N=1, `cpu_baseline` = the reference's own torch field eval timed live on this host's cores (kind "reference": the tree
  is /root/reference in the build container, the byte-compiled oracle/_ref on the GPU box) with `cpu_port_c` (C/OpenMP port
  of the oracle, the parity checker) beside it; without a reference tree: `cpu_baseline` = the C port (kind "port"),
  `reference_torch_cpu` = the figure recorded in the build container, `reference_torch_cpu_port_here` = the torch op
  sequence restated in oracle/torch_cpu_port.py timed on this host.
--config c2: BASELINE config[2], 8192 camera rays through the fused proposal sampler (2 rounds) + fused field/compositing
  with NeuRAD's default grids; roofline on the proposal sampler kernel (192 B per proposal evaluation).
--config c3: the `train_full` step as the timed step.
--config c4: BASELINE config[4], 65536 rays with dynamic actors, appearance embedding and fp16 tables (eval + one training step).
NRHIP_DIST_BACKEND=gloo rehearses the N > 1 code path on a box with fewer GPUs than ranks (labelled, not a measurement).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

# multi-process GPU work on this host driver needs dmabuf IPC (RCCL fails with hipIpcGetMemHandle otherwise); the
# environment exports it already -- kept here for launches that build their own
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

R_RAYS, N_SAMPLES = 4096, 128
GRID = dict(num_levels=16, features_per_level=2, log2_hashmap_size=19, min_res=16, max_res=1024)  # encodings.py:326-333
HIDDEN = 64
STATIC_SCALE = 100.0
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s measured streaming ceiling)
L1_ACCESS_PEAK_G = 256 * 2.4  # vector L1: one cache-line access per clock and CU, 256 CUs x 2.4 GHz = 614 G accesses/s
C3_CAMERA_RAYS, C3_LIDAR_RAYS = 40960, 16384  # ad_datamanager.py:38-41 (40 patches of 32x32 + 16384 lidar points)


def algorithmic_bytes_per_sample(L, F, table_bytes, S):
    """L*8*F*sizeof table reads + 8 B (t_start,t_end) + per-ray I/O (40 B in, 136 B out) / S."""
    return L * 8 * F * table_bytes + 8 + (40 + 136) / S


def make_workload(device, seed):
    from neurad_studio_amd import ops

    g = torch.Generator(device=device)
    g.manual_seed(seed)
    spec = ops.GridSpec(GRID["num_levels"], GRID["features_per_level"], GRID["log2_hashmap_size"], GRID["min_res"],
                        GRID["max_res"])
    table = (torch.rand((spec.table_rows, spec.features_per_level), device=device, generator=g) * 2 - 1) * 1e-3

    def linear(o, i):
        k = 1.0 / np.sqrt(i)
        return ((torch.rand((o, i), device=device, generator=g) * 2 - 1) * k,
                (torch.rand((o,), device=device, generator=g) * 2 - 1) * k)

    H = HIDDEN
    geo = [linear(H, 32), linear(33, H)]
    feat = [linear(H, 48), linear(H, H), linear(32, H)]
    fs = ops.FieldSpec(spec, table, STATIC_SCALE, [w for w, _ in geo], [b for _, b in geo], [w for w, _ in feat],
                       [b for _, b in feat], use_sdf=True, beta=20.0 + 1e-4)
    # SURVEY §8(d) synthetic rays
    origins = torch.randn((R_RAYS, 3), device=device, generator=g) * 5.0
    dirs = torch.randn((R_RAYS, 3), device=device, generator=g)
    dirs = dirs / dirs.norm(dim=-1, keepdim=True)
    area = torch.full((R_RAYS,), 2.7e-7 * 9, device=device)
    fars = torch.full((R_RAYS,), 20000.0, device=device)
    return fs, origins.contiguous(), dirs.contiguous(), area, fars


def recorded_traffic(name):
    """HBM bytes per launch of a kernel from the committed PMC summary profiles/traffic_<name>.json (FETCH_SIZE / WRITE_SIZE
    passes, corrected as MI355X_MICROARCH.md prescribes).  The file carries the sha1 of the kernel sources it was measured
    on (scripts/traffic_stamp.py): when they differ from the tree's, the number is stale and None is reported instead."""
    import hashlib

    tf = os.path.join(ROOT, "profiles", f"traffic_{name}.json")
    if not os.path.exists(tf):
        return None
    rec = json.load(open(tf))
    for rel, want in rec.get("source_sha1", {}).items():
        src = os.path.join(ROOT, "neurad_studio_amd", "csrc", rel)
        if not os.path.exists(src) or hashlib.sha1(open(src, "rb").read()).hexdigest() != want:
            TRAFFIC_NOTES[name] = f"profiles/traffic_{name}.json was measured on another version of {rel}: stale, not reported"
            return None
    return rec.get("hbm_bytes_per_launch")


TRAFFIC_NOTES = {}


LAST_ISSUE = {}


def timed(step, steps, warmup, world, device):
    """the contract's timing: W untimed steps, then exactly K steps between barrier + synchronize, MAX over ranks"""
    import torch.distributed as dist

    for _ in range(warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    LAST_ISSUE["s_per_step"] = (time.perf_counter() - t0) / max(steps, 1)  # host time to ENQUEUE a step (diagnostic)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    return el


def _fused(cls, params, **kw):
    try:
        return cls(params, fused=True, **kw)
    except (RuntimeError, TypeError):
        return cls(params, **kw)


def _split_groups(params, groups):
    """-> {"hashgrids": [...], "fields": [...], "cnn": [...]}: the reference's parameter groups (models/neurad.py:280-289,
    get_param_groups).  Without an explicit assignment: the tables (>= 2^16 elements) are `hashgrids`, the rest `fields`."""
    params = list(params)
    if groups is None:
        return {"hashgrids": [p for p in params if p.numel() >= 1 << 16], "fields": [p for p in params if p.numel() < 1 << 16],
                "cnn": [], "trajectory_opt": []}
    ids = {id(p) for p in params}
    out = {k: [p for p in groups.get(k, []) if id(p) in ids and p.requires_grad] for k in ("hashgrids", "fields", "cnn")}
    taken = {id(p) for v in out.values() for p in v}
    # what no group names: the actor trajectories (the reference's `trajectory_opt`, Adam lr 1e-3: ad_model.py:75-79)
    out["trajectory_opt"] = [p for p in params if id(p) not in taken]
    assert sum(len(v) for v in out.values()) == len(params), "every trained parameter belongs to exactly one group"
    return out


class _Optimizers:
    """the reference's optimizer groups (configs/method_configs.py:415-426): `hashgrids` Adam(lr=1e-2, eps=1e-15) ->
    HashGridAdam (csrc/adam.hip, torch.optim.Adam arithmetic, one streaming kernel per table); `fields` AdamW(lr=1e-2,
    eps=1e-15, weight_decay=1e-7) and `cnn` AdamW(lr=1e-3, eps=1e-15, weight_decay=1e-6) -> torch's fused AdamW"""

    def __init__(self, params, groups=None, capturable=False):
        """capturable: step counts on the device (HashGridAdam's device-controlled form, torch's capturable=True) so that the
        whole step can be captured in a HIP graph"""
        from neurad_studio_amd.optim import HashGridAdam

        g = _split_groups(params, groups)
        kw = {"capturable": True} if capturable else {}
        self.opts = [HashGridAdam(g["hashgrids"], lr=1e-2, eps=1e-15, **kw)] if g["hashgrids"] else []
        if g["fields"]:
            self.opts.append(_fused(torch.optim.AdamW, g["fields"], lr=1e-2, eps=1e-15, weight_decay=1e-7, **kw))
        if g["cnn"]:
            self.opts.append(_fused(torch.optim.AdamW, g["cnn"], lr=1e-3, eps=1e-15, weight_decay=1e-6, **kw))
        if g["trajectory_opt"]:
            self.opts.append(_fused(torch.optim.Adam, g["trajectory_opt"], lr=1e-3, eps=1e-15, **kw))

    def zero_grad(self, set_to_none=True):
        for o in self.opts:
            o.zero_grad(set_to_none=set_to_none)

    def step(self):
        for o in self.opts:
            o.step()


def adam_live_report(opt):
    """csrc/adam.hip reads 12 instead of 28 bytes for elements whose gradient and both moments are exactly zero (rows no ray
    has touched yet -- exact, but a short bench from zero state flatters the kernel: after thousands of steps few rows are
    still untouched).  -> what the timed steps saw and what the kernel costs with EVERY row live: the fraction of table
    elements with a non-zero moment after the run, and the table optimizer's time on copies of the same tensors whose
    moments are all non-zero (HIP events, best of 3)."""
    from neurad_studio_amd import ops
    from neurad_studio_amd.optim import HashGridAdam

    hg = next((o for o in getattr(opt, "opts", []) if isinstance(o, HashGridAdam)), None)
    if hg is None:
        return None
    items, live, total = [], 0, 0
    for group in hg.param_groups:
        for p in group["params"]:
            st = hg.state.get(p)
            if not st:
                continue
            m, v = st["exp_avg"], st["exp_avg_sq"]
            live += int(((m != 0) | (v != 0)).sum())
            total += m.numel()
            tgt = st.get("master", p.detach())
            grad = p.grad.detach().clone() if p.grad is not None else torch.zeros_like(p)  # (cleared after the timed steps)
            items.append((tgt.clone(), grad, torch.full_like(m, 1e-30), torch.full_like(v, 1e-30),
                          int(st["step"]) + 1, None if p.dtype == torch.float32 else p.detach().clone()))
    if not items:
        return None
    g = hg.param_groups[0]
    best = float("inf")
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        big = [it for it in items if it[0].numel() >= 1 << 24]
        for it in big:
            ops.adam_step_many([it], g["lr"], g["betas"][0], g["betas"][1], g["eps"], g["weight_decay"])
        ops.adam_step_many([it for it in items if it[0].numel() < 1 << 24], g["lr"], g["betas"][0], g["betas"][1], g["eps"],
                           g["weight_decay"])
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    nbytes = sum(it[0].numel() * (4 + it[1].element_size() + 4 + 4 + 4 + 4 + (2 if it[5] is not None else 0)) for it in items)
    return {"live_fraction_after_the_timed_steps": live / max(total, 1), "table_elements": total,
            "table_optimizer_ms_all_rows_live": best, "all_live_bytes": nbytes, "all_live_gb_per_s": nbytes / best / 1e6,
            "what": "hash-table Adam (csrc/adam.hip): elements with g = m = v = 0 cost 12 B instead of 28 B; the timed steps start "
                    "from zero state, so most rows are still untouched -- `table_optimizer_ms_all_rows_live` is the same "
                    "launches on the same tables with every moment non-zero (the long-run cost)"}


class _ShardedOptimizers(_Optimizers):
    """--sharded-adam at N > 1: the hash tables on parallel/sharded_adam.py (their reduce-scatter IS the gradient exchange,
    Adam runs on 1/N of each table, the updated parameters are all-gathered); the GradientSynchronizer skips them"""

    def __init__(self, params, groups=None):
        from neurad_studio_amd.parallel.sharded_adam import ShardedTableAdam

        g = _split_groups(params, groups)
        self.tables = g["hashgrids"]
        self.sharded = ShardedTableAdam(self.tables, lr=1e-2, eps=1e-15, usage="static", wire_dtype=WIRE_DTYPE)
        self.opts = [self.sharded]
        if g["fields"]:
            self.opts.append(_fused(torch.optim.AdamW, g["fields"], lr=1e-2, eps=1e-15, weight_decay=1e-7))
        if g["cnn"]:
            self.opts.append(_fused(torch.optim.AdamW, g["cnn"], lr=1e-3, eps=1e-15, weight_decay=1e-6))
        if g["trajectory_opt"]:
            self.opts.append(_fused(torch.optim.Adam, g["trajectory_opt"], lr=1e-3, eps=1e-15))

    def owned_params(self):
        return self.tables


WIRE_DTYPE = None  # --wire-bf16: torch.bfloat16 -> the reduce-scatter leg of the large table gradients in 16 bits

_OPT_GROUPS = ("the reference's groups (configs/method_configs.py:415-426): hashgrids Adam(lr 1e-2, eps 1e-15), fields AdamW(lr 1e-2, "
               "wd 1e-7), cnn AdamW(lr 1e-3, wd 1e-6); ")


def make_optimizer(params, sharded=False, groups=None, capturable=False):
    if capturable and not sharded:
        return _Optimizers(params, groups, capturable=True), (_OPT_GROUPS + "hash tables on nrhip_adam_step_many_dev (dense, "
                                                              "torch.optim.Adam arithmetic, step counts on the device), the rest on "
                                                              "torch's fused AdamW (capturable)")
    if sharded:
        return _ShardedOptimizers(params, groups), (_OPT_GROUPS + "hash tables on ShardedTableAdam (reduce-scatter of the "
                                                    "gradient, Adam on the rank's shard via nrhip_adam_step, all-gather of the "
                                                    "parameters), the rest on torch's fused AdamW")
    return _Optimizers(params, groups), (_OPT_GROUPS + "hash tables on nrhip_adam_step (dense, torch.optim.Adam arithmetic), the "
                                         "rest on torch's fused AdamW")


def train_section(device, rank, world, steps, warmup, beta=None, table_scale=None):
    """train iters/sec on the config-1 workload: PowerSampler bins -> NeuRADField.render_train (ONE autograd node: fused field
    kernel that stores its activations -> learnable-beta SDF head -> weights -> C1/C2 compositing) -> loss -> backward
    (compositing + head incl. d beta in one kernel, MFMA data + weight gradients, radix-partition table gradient without
    memory-side atomics) -> gradient exchange (RCCL reduce-scatter / all-gather on the flat table gradient) -> Adam step.
    beta / table_scale: the non-saturating variant (see main)."""
    from neurad_studio_amd import ops
    from neurad_studio_amd.fields.neurad_field import NeuRADField, NeuRADFieldConfig
    from neurad_studio_amd.parallel.data_parallel import GradientSynchronizer

    torch.manual_seed(7)  # identical replicas on every rank
    cfg = NeuRADFieldConfig(geo_hidden_dim=HIDDEN, nff_hidden_dim=HIDDEN)
    if beta is not None:
        cfg.sdf_beta = beta
    st = cfg.grid.static
    st.num_levels, st.hashgrid_dim, st.log2_hashmap_size = GRID["num_levels"], GRID["features_per_level"], GRID["log2_hashmap_size"]
    st.base_res, st.max_res = GRID["min_res"], GRID["max_res"]
    fld = NeuRADField(cfg, actors=None, static_scale=STATIC_SCALE).to(device).train()
    if table_scale is not None:
        with torch.no_grad():
            t = fld.hashgrid.static_grid.hash_table
            t.mul_(table_scale / float(t.abs().max()))
    fld.order_rays = True  # random rays: the training forward walks them in the nrhip_ray_order order
    opt, opt_name = make_optimizer(fld.parameters())
    sync = GradientSynchronizer(fld.parameters(), average=True, usage="static")
    g = torch.Generator(device=device)
    g.manual_seed(99 + rank)
    o = torch.randn((R_RAYS, 3), device=device, generator=g) * 5.0
    d = torch.randn((R_RAYS, 3), device=device, generator=g)
    d = d / d.norm(dim=-1, keepdim=True)
    target = torch.rand((R_RAYS, 32), device=device, generator=g)
    tdepth = torch.rand((R_RAYS, 1), device=device, generator=g) * 50
    area = torch.full((R_RAYS, 1), 2.43e-6, device=device)
    fars = torch.full((R_RAYS,), 20000.0, device=device)
    state = {}

    def step(_i=None):
        t_rand = torch.rand((R_RAYS, N_SAMPLES + 1), device=device)  # training-mode stratified jitter
        eu = ops.power_sampler(None, fars, N_SAMPLES, -1.0, 0.1, t_rand)[1]
        feats, depth, acc, w = fld.render_train(o, d, area, eu)
        # feature L2 + depth L1 + a weight regulariser, in torch's fused forms (the same three terms as
        # (feats - target).square().mean() + 1e-4 * (depth - tdepth).abs().mean() + 1e-3 * w.square().sum(-1).mean())
        loss = (torch.nn.functional.mse_loss(feats, target) + 1e-4 * torch.nn.functional.l1_loss(depth, tdepth)
                + (1e-3 / R_RAYS) * w.square().sum())
        opt.zero_grad(set_to_none=True)
        loss.backward()
        state["bytes"] = sync.sync()
        opt.step()
        state["loss"] = loss

    el = timed(step, steps, warmup, world, device)
    assert torch.isfinite(state["loss"])
    return {"iters_per_sec": steps / el, "ms_per_iter": el / steps * 1e3, "steps": steps,
            "ray_samples_per_sec": world * R_RAYS * N_SAMPLES * steps / el,
            "grad_exchange_bytes_per_rank": state["bytes"], "grad_exchange_wire_bytes_per_rank": sync.last_wire_bytes,
            "optimizer": opt_name,
            "what": "fwd + bwd + gradient exchange + optimizer step, 4096 rays x 128 samples per GPU"}


def joint_batch(device, rank, n_cam, n_lidar):
    """BASELINE config[3] shape: camera rays as 32x32 pixel patches of a pinhole camera (f=1900 px,
    1920x1280) + lidar rays with is_lidar / did_return / directions_norm metadata, 6 cameras + 1 lidar, 8 s clip."""
    g = torch.Generator(device=device)
    g.manual_seed(4242 + rank)  # every rank draws its own batch (scripts/train.py:104)
    n_p = n_cam // 1024
    cam_o = (torch.randn((n_p, 1, 3), device=device, generator=g) * torch.tensor([30.0, 30.0, 0.5], device=device)
             ).expand(n_p, 1024, 3).reshape(-1, 3)
    fwd = torch.randn((n_p, 3), device=device, generator=g) * torch.tensor([1.0, 1.0, 0.1], device=device)
    fwd = fwd / fwd.norm(dim=-1, keepdim=True)
    up = torch.tensor([0.0, 0.0, 1.0], device=device).expand(n_p, 3)
    right = torch.cross(fwd, up, dim=-1)
    right = right / right.norm(dim=-1, keepdim=True)
    up2 = torch.cross(right, fwd, dim=-1)
    u0 = torch.randint(0, 1920 - 32, (n_p,), device=device, generator=g)
    v0 = torch.randint(0, 1280 - 32, (n_p,), device=device, generator=g)
    vv, uu = torch.meshgrid(torch.arange(32, device=device), torch.arange(32, device=device), indexing="ij")
    x = ((u0[:, None, None] + uu[None]) - 960.0) / 1900.0
    y = ((v0[:, None, None] + vv[None]) - 640.0) / 1900.0
    cam_d = (fwd[:, None, None, :] + x[..., None] * right[:, None, None, :] - y[..., None] * up2[:, None, None, :]).reshape(-1, 3)
    lid_o = (torch.randn((n_lidar, 3), device=device, generator=g) * torch.tensor([30.0, 30.0, 0.3], device=device))
    lid_d = torch.randn((n_lidar, 3), device=device, generator=g) * torch.tensor([1.0, 1.0, 0.15], device=device)
    o = torch.cat([cam_o, lid_o]).contiguous()
    d = torch.cat([cam_d, lid_d])
    d = (d / d.norm(dim=-1, keepdim=True)).contiguous()
    R = n_cam + n_lidar
    is_lidar = (torch.arange(R, device=device) >= n_cam)[:, None]
    area = torch.where(is_lidar, torch.tensor(4.5e-6, device=device), torch.tensor(2.7e-7, device=device))
    md = {"is_lidar": is_lidar, "did_return": torch.rand((R, 1), device=device, generator=g) < 0.8,
          "directions_norm": torch.rand((R, 1), device=device, generator=g) * 78 + 2,
          "sensor_idxs": torch.where(is_lidar, 6, torch.randint(0, 6, (R, 1), device=device, generator=g))}
    times = torch.rand((R, 1), device=device, generator=g) * 8.0
    return o, d, area, times, md


def torch_op_attribution(step_fn, path, n=2):
    """diagnostic (not part of any line): which part of a step / which autograd node launches its torch library kernels
    (fills, adds, copies) -- aten ops with device time of their own over n steps, grouped by the enclosing record_function
    ("S:...") ranges and autograd nodes, with their input shapes"""
    from torch.profiler import ProfilerActivity, profile

    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        for _ in range(n):
            step_fn()
        torch.cuda.synchronize()
    agg = {}
    for ev in prof.events():
        self_us = getattr(ev, "self_device_time_total", None)
        if self_us is None:
            self_us = getattr(ev, "self_cuda_time_total", 0)
        if not ev.name.startswith("aten::") or self_us <= 0:
            continue
        where, p = [], ev.cpu_parent
        while p is not None:
            if p.name.startswith("S:") or "evaluate_function" in p.name or p.name.endswith("Backward") or "Fn" in p.name:
                where.append(p.name.replace("autograd::engine::evaluate_function: ", "bwd:"))
            p = p.cpu_parent
        shapes = str([sh for sh in (getattr(ev, "input_shapes", None) or []) if sh])[:60]
        c = agg.setdefault((ev.name, " <- ".join(where[:3]), shapes), [0, 0.0])
        c[0] += 1
        c[1] += self_us
    # device-to-device copies (hipMemcpyAsync: `__amd_rocclr_copyBuffer` in a kernel trace) are not kernels of their op: they
    # hang off the CPU op that issued them as `kernels` entries named Memcpy / Memset
    mem = {}
    for ev in prof.events():
        for k in (getattr(ev, "kernels", None) or []):
            if "emcpy" not in k.name and "emset" not in k.name:
                continue
            where, p = [], ev.cpu_parent
            while p is not None:
                if p.name.startswith("S:") or "evaluate_function" in p.name or p.name.endswith("Backward") or "Fn" in p.name:
                    where.append(p.name.replace("autograd::engine::evaluate_function: ", "bwd:"))
                p = p.cpu_parent
            shapes = str([sh for sh in (getattr(ev, "input_shapes", None) or []) if sh])[:60]
            c = mem.setdefault((k.name[:28], ev.name, " <- ".join(where[:3]), shapes), [0, 0.0])
            c[0] += 1
            c[1] += k.duration
    with open(path, "w") as f:
        f.write(f"# aten ops with device time of their own over {n} steps: calls, device us, op, enclosing step part / autograd node, input shapes\n")
        for (name, where, shapes), (cnt, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            f.write(f"{cnt:5d} {us:9.1f}  {name:24s} {where}  {shapes}\n")
        f.write(f"# device memcpy / memset nodes over {n} steps: calls, device us, kind, issuing op, enclosing part / node, input shapes\n")
        for (kind, op, where, shapes), (cnt, us) in sorted(mem.items(), key=lambda kv: -kv[1][1]):
            f.write(f"{cnt:5d} {us:9.1f}  {kind:28s} {op:22s} {where}  {shapes}\n")


def train_full_section(device, rank, world, steps, warmup, n_cam=C3_CAMERA_RAYS, n_lidar=C3_LIDAR_RAYS, rgb_decoder=True,
                       cfg_edit=None, sharded_adam=False, sparse_exchange=False, torch_decoder=False, graph=True):
    """The whole training step at the reference's default sizes (models/neurad.py defaults: static grid L=8, F=4, T=2^22;
    proposal grids L=6, F=1, T=2^20; 128+64 proposal samples, 32 field samples; 32-wide MLPs; 16-d appearance embedding;
    lidar head; RGB CNN decoder) on a camera+lidar joint batch: get_nff_outputs (training mode, jitter, lidar metadata) ->
    lidar head + RGB CNN decoder (32x32 feature patches -> 96x96 rgb, models/neurad.py:198-216,359-366; fp16 operands with
    fp32 accumulation like the reference's mixed_precision=True trainer, configs/method_configs.py:401: the HIP kernels of
    csrc/decoder.hip, or with torch_decoder=True the torch modules = MIOpen under fp16 autocast) -> rgb MSE + lidar depth / intensity /
    ray-drop / carving losses + interlevel + distortion (reference multipliers, models/neurad.py:65-94,534-560) -> backward
    -> gradient exchange -> Adam.  rgb_decoder=False: a feature regression stands in for the decoder + rgb loss (round 2's
    step, kept as the `hot path only` variant)."""
    from neurad_studio_amd.cameras.rays import RayBundle
    from neurad_studio_amd.model_components.cnns import decode_rgb, make_rgb_decoder
    from neurad_studio_amd.model_components.lidar_losses import (LidarLossSettings, WeightedLossSum, lidar_loss_multipliers,
                                                                 lidar_metrics, lidar_rows)
    from neurad_studio_amd.model_components.losses import distortion_loss, zipnerf_interlevel_loss
    from neurad_studio_amd.models.neurad import NeuRADHotPath, NeuRADHotPathConfig
    from neurad_studio_amd.parallel.data_parallel import GradientSynchronizer

    torch.manual_seed(11)  # identical replicas
    mcfg = NeuRADHotPathConfig()
    if cfg_edit is not None:
        cfg_edit(mcfg)
    m = NeuRADHotPath(mcfg, static_scale=STATIC_SCALE, num_sensors=7, duration=8.0).to(device).train()
    dec = make_rgb_decoder(mcfg.field.nff_out_dim + mcfg.appearance_dim, 32, mcfg.rgb_upsample_factor).to(device).train() \
        if rgb_decoder else None
    with torch.no_grad():  # O(1) features so that densities / alphas are not degenerate
        m.field.hashgrid.static_grid.hash_table.mul_(1000.0)
        for p in m.proposal_fields:
            p.hashgrid.static_grid.hash_table.mul_(2000.0)
    params = [p for p in m.parameters() if p.requires_grad] + ([] if dec is None else list(dec.parameters()))
    groups = dict(m.get_param_groups(), cnn=[] if dec is None else list(dec.parameters()))
    # graph (one GPU): after the timed eager steps the static step -- no device->host read, no allocation in steady state, step
    # counts and jitter draws on the device -- is also captured in a HIP graph and its replays are timed (`hip_graph_replay`)
    use_graph = bool(graph) and world == 1 and not torch_decoder
    opt, opt_name = make_optimizer(params, sharded=sharded_adam and world > 1, groups=groups, capturable=use_graph)
    # static scene: the used-parameter set is agreed once (no per-step host read); the proposal tables' exchange starts
    # from their gradient hooks, under the field backward
    level_tables = None
    if sparse_exchange and world > 1 and not hasattr(opt, "owned_params"):  # coarse table levels as (row, values) lists
        grids = [m.field.hashgrid.static_grid] + [p.hashgrid.static_grid for p in m.proposal_fields]
        level_tables = {g.hash_table: g.num_levels for g in grids if g.hash_table.dtype == torch.float32}
    sync = GradientSynchronizer(params, average=True, usage="static", overlap=world > 1,
                                skip=opt.owned_params() if hasattr(opt, "owned_params") else (), level_tables=level_tables,
                                profile=world > 1, wire_dtype=WIRE_DTYPE)
    o, d, area, times, md = joint_batch(device, rank, n_cam, n_lidar)
    R = n_cam + n_lidar
    g = torch.Generator(device=device)
    g.manual_seed(5 + rank)
    up = mcfg.rgb_upsample_factor
    image = torch.rand((n_cam // 1024, 32 * up, 32 * up, 3), device=device, generator=g)  # the batch's rgb patches
    target = torch.rand((n_cam, 32 + mcfg.appearance_dim), device=device, generator=g)
    is_lidar = md["is_lidar"][:, 0]
    did_return = md["did_return"][is_lidar][:, 0]
    distance = md["directions_norm"][is_lidar]
    intensity_t = torch.rand((n_lidar, 1), device=device, generator=g)
    lcfg = LidarLossSettings()
    mults = lidar_loss_multipliers(lcfg)
    mults.update(interlevel=0.001, distortion=0.002)  # models/neurad.py:84-85
    mults["rgb" if rgb_decoder else "feature"] = 5.0  # rgb_mult (models/neurad.py:70)
    total_loss = WeightedLossSum(mults, device)
    nears = torch.zeros((R, 1), device=device)
    state = {}

    def decode(cam_features):
        if not torch_decoder:
            return decode_rgb(dec, cam_features, (32, 32))
        with torch.autocast("cuda", dtype=torch.float16):
            return decode_rgb(dec, cam_features, (32, 32), fused=False).float()

    def step(_i=None):
        rb = RayBundle(origins=o, directions=d, pixel_area=area.clone(), nears=nears, fars=None, times=times,
                       metadata=dict(md))
        out = m.get_nff_outputs(rb, calc_lidar_losses=True)
        rows = lidar_rows(is_lidar, n_lidar)  # positions of the lidar rays, no host sync
        out["intensity"], out["ray_drop_logits"] = m.decode_lidar(out["features"], rows=rows[0])
        terms = lidar_metrics(out, is_lidar, did_return, distance, intensity_t, lcfg, rows=rows)
        terms["interlevel"] = zipnerf_interlevel_loss(out["weights_list"], out["ray_samples_list"])
        terms["distortion"] = distortion_loss(out["weights_list"], out["ray_samples_list"])
        if dec is not None:
            terms["rgb"] = torch.nn.functional.mse_loss(decode(out["features"][:n_cam]), image)
        else:
            terms["feature"] = (out["features"][:n_cam] - target).square().mean()
        loss = total_loss(terms)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        state["bytes"] = sync.sync()
        opt.step()
        m.sampler.step_cb(0)
        state["loss"] = loss

    def _device_allocs():
        st = torch.cuda.memory_stats(device)
        return int(st.get("num_device_alloc", 0)), int(st.get("num_alloc_retries", 0))

    for _ in range(warmup):  # the caching allocator reaches its steady state here, outside timed()'s own warm-up
        step()
    torch.cuda.synchronize()
    allocs0 = _device_allocs()
    el = timed(step, steps, 1, world, device)
    allocs1 = _device_allocs()
    host_issue_ms = LAST_ISSUE.get("s_per_step", 0.0) * 1e3
    assert torch.isfinite(state["loss"]), "non-finite loss"
    exchanged_bytes = state["bytes"]
    if os.environ.get("NRHIP_BENCH_TORCH_PROFILE") and rank == 0:
        torch_op_attribution(step, os.environ["NRHIP_BENCH_TORCH_PROFILE"] + ".train_full")
    graph_replay = None
    if use_graph:
        # the same step captured ONCE in a HIP graph (torch's whole-step recipe: the previous iteration's autograd graph is
        # released, a short warm-up runs on a side stream, then the capture) and replayed: what the launch path costs
        try:
            state.clear()
            opt.zero_grad(set_to_none=True)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):
                    step()
            torch.cuda.current_stream().wait_stream(side)
            state.clear()
            opt.zero_grad(set_to_none=True)
            cuda_graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(cuda_graph):
                step()
            n_g = max(steps // 2, 5)
            el_g = timed(lambda _i=None: cuda_graph.replay(), n_g, 2, world, device)
            assert torch.isfinite(state["loss"]), "non-finite loss in the replayed step"
            graph_replay = {"ms_per_iter": el_g / n_g * 1e3, "host_enqueue_ms_per_step": LAST_ISSUE.get("s_per_step", 0.0) * 1e3,
                            "steps": n_g,
                            "what": "the whole step (jitter draws, forward, losses, backward, optimizers with their step counts "
                                    "on the device) captured once with torch.cuda.graph; every step is one hipGraphLaunch.  The "
                                    "eager step above is GPU-bound already (its host enqueue time is below its duration), so "
                                    "the replay cannot be faster than the kernels; where it is slower, that is the graph's "
                                    "per-node dispatch"}
            del cuda_graph
        except Exception as e:  # noqa: BLE001  (a capture failure must not cost the line)
            torch.cuda.synchronize()
            graph_replay = {"error": f"{type(e).__name__}: {e}"[:400]}
        state.clear()
        opt.zero_grad(set_to_none=True)
    s = m.config.sampling
    # roofline of the step's largest single kernel, the fused training forward of the main field (render_kernel storing its
    # activations): timed standalone, after the timed region, on this batch's rays and 32 PowerSampler samples per ray
    from neurad_studio_amd import ops

    S = s.num_nerf_samples
    eu = ops.power_sampler(None, torch.full((R,), s.sky_distance, device=device), S, last_edge=s.sky_distance)[1]
    fspec, a1 = m.field.field_spec(), area.reshape(-1).contiguous()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(8)]
    for k in range(10):
        if k >= 2:
            ev[k - 2][0].record()
        ops.field_fwd_train(fspec, o, d, a1, eu[:, :-1], eu[:, 1:])
        if k >= 2:
            ev[k - 2][1].record()
    torch.cuda.synchronize()
    k_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    dec_ms = None
    if dec is not None:  # the decoder's own share: forward + backward of decode_rgb on this batch's patches, standalone
        f48 = torch.randn((n_cam, 32 + mcfg.appearance_dim), device=device, requires_grad=True)
        evd = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
        for k in range(7):
            if k >= 2:
                evd[k - 2][0].record()
            torch.nn.functional.mse_loss(decode(f48), image).backward()
            if k >= 2:
                evd[k - 2][1].record()
        torch.cuda.synchronize()
        dec_ms = float(np.mean([a.elapsed_time(b) for a, b in evd]))
        opt.zero_grad(set_to_none=True)
    g = m.field.hashgrid.static_grid
    H = m.config.field.geo_hidden_dim
    tb = 2 if g.hash_table.dtype == torch.float16 else 4
    per_sample = algorithmic_bytes_per_sample(g.num_levels, g.features_per_level, tb, S) + 4 * (32 + H + 48 + 2 * H) + 4 * 34
    roof = {"kernel": f"nrhip::render_kernel<{g.num_levels},{g.features_per_level},{H},{'fp16' if tb == 2 else 'fp32'},train> "
                      "(fused field forward that stores its activations; timed standalone after the step loop)",
            "bound": "hbm", "achieved": R * S * per_sample / (k_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": R * S * per_sample / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "traffic": recorded_traffic("field_fwd_train") if (n_cam, n_lidar, tb) == (C3_CAMERA_RAYS, C3_LIDAR_RAYS, 4)
            else None,
            "algorithmic_bytes_per_launch": R * S * per_sample, "kernel_ms": k_ms,
            "bytes_per_sample": "table reads L*8*F*sizeof(entry) + 8 B interval + per-ray I/O / S + saved activations "
                                "(32 + H + 48 + 2H floats) + per-sample outputs (34 floats)"}
    return {"roofline": roof, "iters_per_sec": steps / el, "ms_per_iter": el / steps * 1e3, "steps": steps,
            "rays_per_sec": world * R * steps / el, "rays_per_gpu": R, "camera_rays": n_cam, "lidar_rays": n_lidar,
            "field_samples_per_ray": s.num_nerf_samples, "proposal_samples_per_ray": list(s.num_proposal_samples),
            "host_enqueue_ms_per_step": host_issue_ms,  # Python + autograd + ctypes time to issue a step, GPU not waited for
            "hip_graph_replay": graph_replay,
            "device_allocations_during_the_timed_steps": allocs1[0] - allocs0[0],  # hipMalloc calls: 0 in steady state
            "allocator_retries_during_the_timed_steps": allocs1[1] - allocs0[1],
            "grad_exchange_bytes_per_rank": exchanged_bytes, "grad_exchange_wire_bytes_per_rank": sync.last_wire_bytes,
            "grad_exchange": ("level-sparse: table levels sent as (row, values) lists this step, by parameter index: "
                              f"{sync.last_list_levels}") if level_tables else
            ("dense; reduce-scatter leg in bf16 (rounded once per rank, all-to-all to the owner, fp32 sum), fp32 all-gather"
             if WIRE_DTYPE is not None else "dense reduce-scatter + all-gather"),
            # N > 1: what the exchange costs the step and how much of it the hooks hid (GradientSynchronizer.timing)
            "grad_exchange_timing": dict(sync.timing(last=steps),
                                         wire_bytes_by_table={("small" if i < 0 else f"param{i}:{tuple(sync.params[i].shape)}"): b
                                                              for i, b in sync.last_wire_bytes_by_param.items()}) if world > 1 else None,
            "optimizer": opt_name, "table_optimizer_live_rows": adam_live_report(opt),
            "rgb_decoder": ("CNN decoder (4 BasicBlocks of 7x7 convs + BatchNorm, 3x transposed conv) + rgb MSE in the step, "
                            + ("torch modules = MIOpen under fp16 autocast" if torch_decoder else
                               "HIP kernels (csrc/decoder.hip: fp16 operands, fp32 accumulation, v_mfma_f32_32x32x16_f16)")
                            + f"; standalone forward+backward {dec_ms:.2f} ms") if dec is not None
            else "not in this step (feature regression stands in)",
            "rgb_decoder_fwd_bwd_ms": dec_ms,
            "what": "BASELINE config[3] shape per GPU: NeuRAD-default grids, sampler (2 rounds) + field + compositing + "
                    "appearance + lidar head" + (" + RGB CNN decoder" if dec is not None else "") +
                    " + rgb/lidar/interlevel/distortion losses, backward, gradient exchange, optimizer step.  NOT in it: the "
                    "VGG perceptual term of the reference's step (models/neurad.py:260,538: a torchvision network outside the hot "
                    "path, SURVEY §8) -- so this figure must not be set beside the reference's it/s of its full trainer"}


def train_via_plugin_section(device, rank, world, steps, warmup, n_cam=C3_CAMERA_RAYS, n_lidar=C3_LIDAR_RAYS, mixed_precision=True,
                             fused_losses=True, table_dtype="float32"):
    """The c3 training step as `ns-train neurad-hip` runs it (--via-plugin): the reference's OWN iteration --
    ``Trainer.train_iteration`` (engine/trainer.py:535-579: zero_grad_some -> torch.autocast -> pipeline.get_train_loss_dict ->
    grad_scaler.scale(loss).backward() -> optimizer_scaler_step_some(grad_scaler) -> grad_scaler.update() -> schedulers) --
    over ``NeuRADHipModel`` (integration/neurad_hip.py, a subclass of the reference's NeuRADModel: its get_outputs,
    decode_features, get_loss_dict are the reference's code), the reference's ``Optimizers`` built from the method's own
    optimizer table (HashGridAdam for `hashgrids` through GradScaler's device-side protocol, torch AdamW / Adam for the rest,
    the reference's schedulers) and ``ADHipPipeline.get_train_loss_dict``; mixed precision on, as the `neurad` method ships.
    Same batch, same sizes, same loss terms as `train_full` (no VGG term in either).  Needs a neurad-studio installation:
    `nerfstudio` must be importable -- on the bench box the only copy is the byte-compiled tree under oracle/_ref, used here
    as the HOST framework the plugin subclasses (never as a checker, and nothing of it is the thing measured apart from its
    own Python glue, which is the point of this figure)."""
    import types
    from collections import defaultdict
    from copy import deepcopy

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle"))
    import ref_import

    if not ref_import.reference_available():
        return {"error": "nerfstudio is not importable (no /root/reference, no oracle/_ref): --via-plugin needs neurad-studio"}
    ref_import.install()
    os.environ.setdefault("NERFSTUDIO_METHOD_CONFIGS", "neurad-hip=neurad_studio_amd.integration.neurad_hip:neurad_hip")
    import nerfstudio.configs.method_configs as ref_methods
    import nerfstudio.models.neurad as ref_neurad
    from nerfstudio.cameras.rays import RayBundle
    from nerfstudio.data.scene_box import SceneBox
    from nerfstudio.engine.optimizers import Optimizers
    from nerfstudio.engine.trainer import Trainer
    from nerfstudio.plugins.registry import discover_methods
    from torch.cuda.amp.grad_scaler import GradScaler

    from neurad_studio_amd.integration.pipeline import ADHipPipeline
    from neurad_studio_amd.optim import TableGradScaler
    from neurad_studio_amd.parallel.data_parallel import GradientSynchronizer

    methods = dict(ref_methods.all_methods)
    if "neurad-hip" not in methods:
        methods.update(discover_methods()[0])
    method = deepcopy(methods["neurad-hip"])
    mcfg = method.pipeline.model
    mcfg.loss.vgg_mult = 0.0  # (the VGG term needs torchvision's weights; train_full leaves it out too)
    mcfg.fused_losses, mcfg.table_dtype = fused_losses, table_dtype
    ref_neurad.VGGPerceptualLossPix2Pix = torch.nn.Identity
    torch.manual_seed(11)  # identical replicas
    m = mcfg.setup(scene_box=SceneBox(aabb=torch.tensor([[-STATIC_SCALE] * 3, [STATIC_SCALE] * 3])), num_train_data=7,
                   metadata={"duration": 8.0, "sensor_idx_to_name": {i: f"s{i}" for i in range(7)}, "trajectories": []})
    m = m.to(device).train()
    m.psnr = lambda x, y: -10.0 * torch.log10(torch.nn.functional.mse_loss(x, y))  # (torchmetrics is absent on this box)
    with torch.no_grad():  # O(1) features, as in train_full
        m.field.hashgrid.static_grid.hash_table.mul_(1000.0)
        for p in m.proposal_fields:
            p.hashgrid.static_grid.hash_table.mul_(2000.0)
    o, d, area, times, md = joint_batch(device, rank, n_cam, n_lidar)
    g = torch.Generator(device=device)
    g.manual_seed(5 + rank)
    up = mcfg.rgb_upsample_factor
    image = torch.rand((n_cam // 1024, 32 * up, 32 * up, 3), device=device, generator=g)
    torch.rand((n_cam, 48), device=device, generator=g)  # (train_full's feature target: the same generator stream after it)
    is_lidar = md["is_lidar"]
    points = torch.cat([torch.zeros((n_lidar, 3), device=device), torch.rand((n_lidar, 1), device=device, generator=g),
                        torch.zeros((n_lidar, 1), device=device)], -1)
    batch = {"image": image, "lidar": points, "is_lidar": is_lidar, "did_return": md["did_return"],
             "distance": md["directions_norm"][is_lidar[:, 0]].reshape(-1, 1).contiguous()}
    cam_idx = torch.zeros((n_cam + n_lidar, 1), dtype=torch.long, device=device)

    def next_train(step):
        rb = RayBundle(origins=o, directions=d, pixel_area=area.clone(), camera_indices=cam_idx, times=times,
                       metadata=dict(md))
        return rb, dict(batch)

    pipe = types.SimpleNamespace(_model=m, model=m, datamanager=types.SimpleNamespace(next_train=next_train),
                                 config=types.SimpleNamespace(ray_patch_size=(32, 32)))
    pipe.get_train_loss_dict = lambda step: ADHipPipeline.get_train_loss_dict(pipe, step)
    groups = {k: v for k, v in m.get_param_groups().items() if len(v)}
    # the trainer class the method's config names (integration/trainer.py: HipTrainer, the reference's iteration with the
    # schedulers' step deferred past its two get_scale() host reads), given the attributes Trainer.__init__ / setup set
    trainer_cls = method._target
    loop = object.__new__(trainer_cls)
    loop.__dict__.update(config=types.SimpleNamespace(log_gradients=False, deferred_scheduler_step=True),
                         device=f"cuda:{device.index or 0}", mixed_precision=bool(mixed_precision),
                         grad_scaler=TableGradScaler(enabled=bool(mixed_precision)),  # (HipTrainer.__init__)
                         gradient_accumulation_steps=defaultdict(lambda: 1), pipeline=pipe,
                         optimizers=Optimizers(deepcopy({k: method.optimizers[k] for k in groups}), groups))
    params = [p for p in m.parameters() if p.requires_grad]
    sync = GradientSynchronizer(params, average=True, usage="static", overlap=world > 1, auto_sync=True,
                                wire_dtype=WIRE_DTYPE) if world > 1 else None
    state = {}

    def step(i=None, iterate=trainer_cls.train_iteration):
        loss, loss_dict, _ = iterate(loop, 0 if i is None else i)
        m.sampler.step_cb(0)  # the model's AFTER_TRAIN_ITERATION callback (models/neurad.py:291-300)
        state["loss"] = loss

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    el = timed(step, steps, 1, world, device)
    host_issue_ms = LAST_ISSUE.get("s_per_step", 0.0) * 1e3
    assert torch.isfinite(state["loss"]), "non-finite loss"
    # the same model, optimizers and scaler stepped by the reference's own Trainer.train_iteration (its two get_scale() reads)
    loop._settle_schedulers()
    hip_scaler, loop.grad_scaler = loop.grad_scaler, GradScaler(enabled=bool(mixed_precision))  # (engine/trainer.py:189)
    loop.grad_scaler.load_state_dict(hip_scaler.state_dict())
    el_ref = timed(lambda i=None: step(i, Trainer.train_iteration), steps, 1, world, device)
    if os.environ.get("NRHIP_BENCH_TORCH_PROFILE") and rank == 0:
        torch_op_attribution(step, os.environ["NRHIP_BENCH_TORCH_PROFILE"] + ".via_plugin")
    return {"iters_per_sec": steps / el, "ms_per_iter": el / steps * 1e3, "steps": steps,
            "rays_per_sec": world * (n_cam + n_lidar) * steps / el, "host_enqueue_ms_per_step": host_issue_ms,
            "trainer": trainer_cls.__name__, "ms_per_iter_under_the_reference_trainer": el_ref / steps * 1e3,
            "grad_scaler": type(hip_scaler).__name__,
            "mixed_precision": bool(mixed_precision), "grad_scaler_scale": loop.grad_scaler.get_scale() if mixed_precision else None,
            "fused_losses": bool(fused_losses), "table_dtype": table_dtype,
            "optimizers": {k: type(v).__name__ for k, v in loop.optimizers.optimizers.items()},
            "grad_exchange_bytes_per_rank": sync.last_sync_bytes if sync is not None else 0,
            "what": "the c3 step as `ns-train neurad-hip` executes it: the method's trainer class (HipTrainer = the reference's "
                    "Trainer.train_iteration -- autocast + GradScaler + Optimizers + schedulers -- with the schedulers' step "
                    "deferred to the next iteration's optimizer step instead of two get_scale() host reads, and a GradScaler "
                    "whose inf check over the table gradients is read-only) over NeuRADHipModel "
                    "/ ADHipPipeline.get_train_loss_dict with the method's own optimizer table; same batch, sizes and loss "
                    "terms as train_full; ms_per_iter_under_the_reference_trainer = the same objects stepped by the "
                    "reference's Trainer.train_iteration itself with torch's GradScaler"}


def device_state(device_index=0):
    """clocks / power state of the GPU UNDER LOAD when the run starts (rocm-smi sampled while a matmul loop keeps the device
    busy: the idle sclk says nothing): the pool's boxes differ by ~10 % on the same kernel (round 3: 169 us on the builder's
    best box, 185 us on the driver's) -- with this in the line the spread is attributable"""
    import subprocess

    try:
        p = subprocess.Popen(["rocm-smi", "-d", str(device_index), "--showclocks", "--showpower", "--showmaxpower",
                              "--showperflevel", "--json"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
        a = torch.randn((4096, 4096), device=f"cuda:{device_index}", dtype=torch.float16)
        t0 = time.perf_counter()
        while p.poll() is None and time.perf_counter() - t0 < 20:
            for _ in range(20):
                a @ a
            torch.cuda.synchronize()
        out = p.communicate(timeout=5)[0]
        d = json.loads(out)
        card = d.get(f"card{device_index}", next(iter(d.values())))
        keep = {"sampled": "under a matmul load"}
        for k, v in card.items():
            kl = k.lower()
            if any(t in kl for t in ("sclk", "mclk", "fclk", "power", "performance level")):
                keep[k] = v
        return keep
    except Exception as e:  # noqa: BLE001
        return {"error": f"{type(e).__name__}: {e}"[:200]}


def cpu_baseline(fs, origins, dirs, area, edges, budget_s=12.0):
    """The oracle's C/OpenMP port on the host cores, on a bounded slice of the SAME workload."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import neurad_oracle as O
    import oracle_c

    h = lambda t: t.detach().cpu().numpy()  # noqa: E731
    grid = O.GridParams(h(fs.table), GRID["num_levels"], GRID["min_res"], GRID["max_res"], GRID["log2_hashmap_size"])
    p = O.FieldParams(grid, STATIC_SCALE, [h(w) for w in fs.geo_w], [h(b) for b in fs.geo_b], [h(w) for w in fs.feat_w],
                      [h(b) for b in fs.feat_b], beta=20.0, use_sdf=True)
    o, d, a, e = h(origins), h(dirs), h(area), h(edges)
    s0, e0 = np.ascontiguousarray(e[:, :-1]), np.ascontiguousarray(e[:, 1:])

    def run(n):
        t0 = time.perf_counter()
        out = oracle_c.render_fwd(p, o[:n], d[:n], a[:n], s0[:n], e0[:n])
        return time.perf_counter() - t0, out

    run(256)  # warm (page-in of the 64 MB table, OpenMP pool)
    _, out = run(R_RAYS)
    reps, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s:  # bounded: ~budget_s seconds of CPU work
        run(R_RAYS)
        reps += 1
    t = time.perf_counter() - t0
    return {"value": reps * R_RAYS * N_SAMPLES / t, "unit": "ray-samples/s", "cores": oracle_c.num_threads(),
            "kind": "port",
            "sample": f"{reps} passes over the full bench batch ({R_RAYS} rays x {N_SAMPLES} samples), {t:.1f} s of "
                      "oracle/neurad_oracle_c.c (C + OpenMP on all host cores, fp32)"}, (R_RAYS, out)


def torch_port_cpu(fs, origins, dirs, area, edges, budget_s=10.0):
    """The reference's field-eval path as torch ops on THIS host's cores (oracle/torch_cpu_port.py: the op sequence of
    HashEncoding.pytorch_fwd + MLP.pytorch_fwd + the dense compositing, pinned to the numpy oracle by
    tests/test_oracle_torch_port.py), on a bounded slice of the bench batch.  What the GPU box can time of "the reference's
    CPU PyTorch path": the reference tree itself is not there."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import neurad_oracle as O
    import torch_cpu_port as P

    h = lambda t: t.detach().cpu().numpy()  # noqa: E731
    grid = O.GridParams(h(fs.table), GRID["num_levels"], GRID["min_res"], GRID["max_res"], GRID["log2_hashmap_size"])
    p = O.FieldParams(grid, STATIC_SCALE, [h(w) for w in fs.geo_w], [h(b) for b in fs.geo_b], [h(w) for w in fs.feat_w],
                      [h(b) for b in fs.feat_b], beta=20.0, use_sdf=True)
    n = 512  # rays per pass (x 128 samples): ~1 s of torch CPU work
    o, d, a, e = (t.detach().cpu()[:n].contiguous() for t in (origins, dirs, area, edges))
    s0, e0 = e[:, :-1].contiguous(), e[:, 1:].contiguous()
    P.render_rays(p, o, d, a, s0, e0)  # warm: thread pool, page-in of the table
    reps, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        P.render_rays(p, o, d, a, s0, e0)
        reps += 1
    t = time.perf_counter() - t0
    return {"value": reps * n * N_SAMPLES / t, "unit": "ray-samples/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{reps} passes over {n} rays x {N_SAMPLES} samples of the bench batch, {t:.1f} s",
            "what": "oracle/torch_cpu_port.py: the reference's torch formulation of the field eval + dense compositing "
                    "(forward, fp32, torch CPU ops on all host threads), live on this host"}


def reference_torch_cpu():
    """The reference's OWN torch field-eval path (NeuRADField(implementation="torch"): fields/neurad_field.py:128-152,
    field_components/encodings.py:406-466) timed LIVE on this host's cores, ``torch.get_num_threads()`` threads
    (oracle/time_reference_cpu.py in a subprocess, ~30 s).  The tree it imports: /root/reference in the build container,
    oracle/_ref (the reference byte-compiled by oracle/make_ref.py; ships with the lease) on the GPU box.  Only where
    neither exists the figure recorded in the build container (profiles/reference_torch_cpu.json) is quoted, labelled."""
    import subprocess

    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ref_import

    note = ""
    if ref_import.reference_available():
        try:
            res = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "time_reference_cpu.py"), "--no-write"],
                                 capture_output=True, text=True, timeout=420)
            r = json.loads(res.stdout.strip().splitlines()[-1])
            return {"value": r["forward_ray_samples_per_s"], "unit": "ray-samples/s", "cores": r["torch_threads"],
                    "host_cpus": r["cores"], "kind": "reference",
                    "forward_backward_value": r["forward_backward_ray_samples_per_s"],
                    "sample": f"median forward pass over {r['rays']} rays x {r['samples']} samples of the bench workload "
                              f"({r['forward_s']:.2f} s) at the best of the thread counts tried "
                              f"({r['forward_s_by_threads']} s per pass; torch's default here: {r['torch_threads_default']}); "
                              f"forward+backward {r['forward_backward_s']:.2f} s per pass",
                    "where": f"this host, live ({ref_import.reference_kind()} tree): " + r["what"]
                             + "; oracle/time_reference_cpu.py"}
        except Exception as e:  # noqa: BLE001  (fall back to the recorded figure, say why)
            note = f"live timing failed ({type(e).__name__}: {str(e)[:120]}); "
    f = os.path.join(ROOT, "profiles", "reference_torch_cpu.json")
    if not os.path.exists(f):
        return None
    r = json.load(open(f))
    return {"value": r["forward_ray_samples_per_s"], "unit": "ray-samples/s", "cores": r["cores"], "kind": "reference",
            "forward_backward_value": r["forward_backward_ray_samples_per_s"],
            "where": note + "build container (no reference tree on this host): " + r["what"] + "; oracle/time_reference_cpu.py"}


def bench_c1(args, device, rank, world):
    from neurad_studio_amd import ops

    fs, origins, dirs, area, fars = make_workload(device, seed=1234 + rank)  # seed + rank like scripts/train.py:104
    S = N_SAMPLES
    feats = torch.empty((R_RAYS, 32), device=device)
    depth = torch.empty((R_RAYS, 1), device=device)
    acc = torch.empty((R_RAYS, 1), device=device)
    state = {}
    # HIP events around the render kernel on every 4th timed step (each record is a marker packet in the queue: on every
    # step they cost ~1.5 % of it)
    ev_every = 4
    events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
              for _ in range((args.steps + ev_every - 1) // ev_every)]

    def step(i=None):
        # M1 sky stretch (models/neurad.py:451-455) folded into the sampler launch; far == sky_distance here anyway
        # bins + the processing order of this batch (cache-locality hint, csrc/rayorder.h): part of the step, recomputed
        # every time, ONE launch (workgroup 0 sorts while the others fill bins).  (As two launches the step was 9 us
        # longer; the ordering pass on a second stream next to the sampler was slower still: 0.206 vs 0.190 ms per step --
        # the two event waits cost more than the kernel.)
        sp, eu, order = ops.power_sampler_ordered(None, fars, S, origins, dirs, STATIC_SCALE, lam=-1.0, scaling=0.1,
                                                  last_edge=20000.0)
        timed_kernel = i is not None and i % ev_every == 0
        if timed_kernel:
            events[i // ev_every][0].record()
        ops.render_fwd(fs, origins, dirs, area, eu[:, :-1], eu[:, 1:], out=(feats, depth, acc), order=order)
        if timed_kernel:
            events[i // ev_every][1].record()
        state["edges"], state["order"] = eu, order

    elapsed = timed(step, args.steps, args.warmup, world, device)
    assert torch.isfinite(feats).all() and torch.isfinite(acc).all()
    n_samples = R_RAYS * S
    out = None
    if rank == 0:
        # Events bracket every 4th launch, starting with the first of the timed region -- and that one is not like the others:
        # it follows the contract's synchronize (a system-scope release: the table's lines leave the L2) and runs ~170 us
        # against ~142 us for every later launch.  Sampled 1 in 4 it would weigh 4 x what it does among the K launches; the
        # average launch duration over the timed region weighs it 1 / K.
        ts = [a.elapsed_time(b) for a, b in events]
        kernel_ms = float(ts[0] if len(ts) == 1 else (ts[0] + (args.steps - 1) * np.mean(ts[1:])) / args.steps)
        if os.environ.get("NRHIP_BENCH_DUMP_EVENTS"):
            print("kernel us per timed event:", [round(t * 1e3, 1) for t in ts], file=sys.stderr)
        bytes_per = algorithmic_bytes_per_sample(GRID["num_levels"], GRID["features_per_level"], 4, S)
        achieved = n_samples * bytes_per / (kernel_ms * 1e-3) / 1e9
        traffic = recorded_traffic("render_kernel")
        out = {
            "metric": "ray-samples/sec (4096 rays x 128 samples)", "value": world * n_samples * args.steps / elapsed,
            "unit": "ray-samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE config[1]: NeuRAD hash-grid (16 levels, T=2^19, F=2, fp32) + 64-wide MLPs, "
                                   "4096 rays x 128 samples per GPU, PowerSampler bins + ray ordering pass + fused field + "
                                   "compositing (forward / render path)",
                       "rays_per_gpu": R_RAYS, "samples_per_ray": S, "parallelism": f"rays sharded x{world}, no collective"},
            "per_gpu_value": n_samples * args.steps / elapsed,
            "target_per_gpu": 2e7,
            "roofline": {"kernel": "nrhip::render_kernel<16,2,64,fp32,composite,pairs> (software-pipelined gathers, "
                                   "XCD-coherent ray ranges over the nrhip_ray_order permutation, MLP products as fp16 "
                                   "pairs with fp32 accumulation)", "bound": "hbm",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "algorithmic_bytes_per_launch": n_samples * bytes_per,
                         "kernel_ms": kernel_ms, "kernel_ms_plain_mean": float(np.mean(ts)),
                         "frac_at_the_plain_mean": n_samples * bytes_per / (float(np.mean(ts)) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "kernel_ms_how": f"HIP events around every 4th of the {args.steps} timed launches; the first one (right after "
                                          f"the synchronize, {ts[0] * 1e3:.1f} us) weighted 1 / {args.steps}, the others' mean "
                                          f"({float(np.mean(ts[1:]) if len(ts) > 1 else ts[0]) * 1e3:.1f} us) {args.steps - 1} / {args.steps}"},
        }
        # two NON-headline variants of the same kernel on the same batch (labelled; the headline stays fp32 / exact):
        # fp16 table storage (BASELINE config 5's layout) and eval-time early ray termination at transmittance 1e-4
        def kernel_us(fspec, **kw):
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
            for k in range(13):
                if k >= 3:
                    evs[k - 3][0].record()
                ops.render_fwd(fspec, origins, dirs, area, state["edges"][:, :-1], state["edges"][:, 1:], order=state["order"], **kw)
                if k >= 3:
                    evs[k - 3][1].record()
            torch.cuda.synchronize()
            return float(np.mean([a.elapsed_time(b) for a, b in evs])) * 1e3

        import dataclasses

        fs16 = dataclasses.replace(fs, table=fs.table.half())
        if os.environ.get("NRHIP_MLP_PAIRS", "1") == "1" and not args.no_variants:
            # the same launch with the matrix products on the fp32 MFMA (the default up to round 4), and how far the two
            # forms' outputs are apart on this batch (both are fp32 product sums to rounding: csrc/render.hip)
            pair_us = kernel_us(fs)
            f_pairs = ops.render_fwd(fs, origins, dirs, area, state["edges"][:, :-1], state["edges"][:, 1:], order=state["order"])
            os.environ["NRHIP_MLP_PAIRS"] = "0"
            ops.reload_tuning()  # (the library reads its switches at load: csrc/common.h struct Tuning)
            try:
                f32_us = kernel_us(fs)
                f_f32 = ops.render_fwd(fs, origins, dirs, area, state["edges"][:, :-1], state["edges"][:, 1:], order=state["order"])
            finally:
                del os.environ["NRHIP_MLP_PAIRS"]
                ops.reload_tuning()
            rl2 = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())  # noqa: E731
            out["mlp_products"] = {
                "form": "fp16 pairs: x = fp16(x) + fp16(x - fp16(x)), three v_mfma_f32_16x16x32_f16 terms per 32 inputs, fp32 "
                        "accumulation; tile in units of 2^6, weights staged x 2^7; tiles / weights that do not fit take the "
                        "fp32 MFMA (NRHIP_MLP_PAIRS=0: the fp32 MFMA everywhere)",
                "kernel_us_back_to_back": {"fp16_pairs": pair_us, "fp32_mfma": f32_us},
                "rel_l2_pairs_vs_fp32_mfma": {k: rl2(a, b) for k, a, b in zip(("features", "depth", "accumulation"), f_pairs, f_f32)}}
        if not args.no_variants:
            out["variants_not_headline"] = {
                "fp16_table_kernel_us": kernel_us(fs16),
                "early_stop_eps_1e-4_kernel_us": kernel_us(fs, early_stop_eps=1e-4),
                "what": "render_kernel on the headline batch with (a) the hash table stored as fp16 (half the gather bytes, "
                        "arithmetic unchanged) and (b) rays stopped once their transmittance is below 1e-4 (error bounded "
                        "by it); neither is the headline configuration"}
    return out, (fs, origins, dirs, area, state["edges"], feats)


def bench_c2(args, device, rank, world):
    """BASELINE config[2]: 8192 camera rays, fused proposal sampler (2 rounds, 128+64 proposal evaluations per ray) +
    fused field/compositing (32 samples), NeuRAD default grids (static L=8,F=4,T=2^22; proposals L=6,F=1,T=2^20)."""
    from neurad_studio_amd.cameras.rays import RayBundle
    from neurad_studio_amd.models.neurad import NeuRADHotPath, NeuRADHotPathConfig

    R = 8192
    torch.manual_seed(3)
    m = NeuRADHotPath(NeuRADHotPathConfig(appearance_dim=0, lidar_decoder=False), static_scale=STATIC_SCALE).to(device).eval()
    with torch.no_grad():
        m.field.hashgrid.static_grid.hash_table.mul_(300.0)
        for p in m.proposal_fields:
            p.hashgrid.static_grid.hash_table.mul_(1000.0)
    o, d, area, times, md = joint_batch(device, rank, R, 0)
    pf = [m.proposal_fields[-1]] * 2
    sky = m.config.sampling.sky_distance
    fars = torch.full((R, 1), sky, device=device)
    nears = torch.zeros((R, 1), device=device)
    ev_every = 4  # kernel-timing events on every 4th timed step (each record is a marker packet in the queue)
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range((args.steps + ev_every - 1) // ev_every)]
    state = {}
    area9 = area * 9.0  # _scale_pixel_area (models/neurad.py:702-709) of a fixed batch, once

    @torch.no_grad()
    def step(i=None):
        rb = RayBundle(origins=o, directions=d, pixel_area=area9, nears=nears, fars=fars)
        e = ev[i // ev_every] if i is not None and i % ev_every == 0 else None
        if e is not None:
            e[0].record()
        rs, pw, prs = m.sampler.generate_fused(rb, pf, sky)
        if e is not None:
            e[1].record()
        fr = rs.frustums
        ends = fr.ends[..., 0].clone()
        ends[:, -1] = sky
        state["out"] = m.field.render(o, d, rb.pixel_area, fr.starts[..., 0], ends)
        if e is not None:
            e[2].record()

    elapsed = timed(step, args.steps, args.warmup, world, device)
    assert torch.isfinite(state["out"][0]).all()
    if rank != 0:
        return None
    s = m.config.sampling
    n_prop, n_field = R * sum(s.num_proposal_samples), R * s.num_nerf_samples
    t_samp = float(np.mean([e[0].elapsed_time(e[1]) for e in ev]))
    t_rend = float(np.mean([e[1].elapsed_time(e[2]) for e in ev]))
    prop_bytes = n_prop * 6 * 8 * 1 * 4  # 192 B per proposal evaluation
    # The kernel's 48 four-byte gathers per evaluation hit in L1 / L2 (0.19 M fabric reads per launch, the tables'
    # compulsory footprint): it runs against the L1's tag-lookup rate -- one cache-line access per clock and CU --, not
    # against HBM (profiles/r02_c2_sampler_pmc.txt: 63.7 M TCP accesses per launch).
    gathers = n_prop * 6 * 8
    l1_peak = L1_ACCESS_PEAK_G
    ach = gathers / (t_samp * 1e-3) / 1e9
    out = {
        "metric": "ray-samples/sec (8192 camera rays, proposal sampler 128+64 -> 32 field samples)",
        "value": world * n_field * args.steps / elapsed, "unit": "ray-samples/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BASELINE config[2]: 8192 camera rays (8 patches of 32x32), fused proposal sampler (2 rounds) + "
                               "fused field + compositing, NeuRAD default grids, eval", "rays_per_gpu": R,
                   "parallelism": f"rays sharded x{world}, no collective"},
        "rays_per_sec": world * R * args.steps / elapsed, "proposal_evals_per_sec": world * n_prop * args.steps / elapsed,
        "roofline": {"kernel": "nrhip::proposal_sampler_kernel (all rounds on chip, one wave per ray)", "bound": "l1",
                     "achieved": ach, "peak": l1_peak, "unit": "Gaccess/s", "frac": ach / l1_peak,
                     "what": "algorithmic 4-byte table gathers (48 per proposal evaluation) per second against the vector "
                             "L1's tag-lookup peak, 256 CUs x 1 line access per clock x 2.4 GHz",
                     "traffic": recorded_traffic("proposal_sampler"), "algorithmic_bytes_per_launch": prop_bytes,
                     "hbm_frac_of_peak": prop_bytes / (t_samp * 1e-3) / 1e9 / HBM_PEAK_GBS, "kernel_ms": t_samp},
        "render_kernel_ms": t_rend,
    }
    if world == 1:  # the whole chain of a 512-ray slice against the C restatement of the oracle
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import neurad_oracle as O
        import oracle_c

        n = 512
        h = lambda t: t.detach().cpu().numpy()  # noqa: E731
        props, fp = _oracle_params(m, O)
        so = oracle_c.proposal_sampler(props, h(o[:n]), h(d[:n]), h(area9[:n]).reshape(-1), np.zeros(n, np.float32),
                                       np.full(n, sky, np.float32))
        ref = oracle_c.render_fwd(fp, h(o[:n]), h(d[:n]), h(area9[:n]).reshape(-1), so["starts"], so["ends"])
        feats = h(state["out"][0][:n])
        out["parity_rel_l2_vs_oracle"] = {
            "features": float(np.linalg.norm(feats - ref["features"]) / np.linalg.norm(ref["features"])),
            "accumulation": float(np.linalg.norm(h(state["out"][2][:n]) - ref["accumulation"]) / np.linalg.norm(ref["accumulation"])),
            "tolerance": 1e-4, "sample": f"{n} rays through oracle/neurad_oracle_c.c: proposal sampler (2 rounds) + field + "
                                         "compositing, the step's own rays and parameters",
            "compositing": "unpinned against nerfacc itself (DESIGN.md §3)"}
    return out


def _oracle_params(m, O):
    """the hot-path model's parameters as the oracle's dataclasses"""
    h = lambda t: t.detach().float().cpu().numpy()  # noqa: E731
    pg, fg = m.proposal_fields[0].hashgrid.config.static, m.field.hashgrid.config.static
    props = [O.ProposalParams(O.GridParams(h(p.hashgrid.static_grid.hash_table), pg.num_levels, pg.base_res, pg.max_res,
                                           pg.log2_hashmap_size), STATIC_SCALE, h(p.density_decoder.weight))
             for p in m.proposal_fields]
    f = m.field
    fp = O.FieldParams(O.GridParams(h(f.hashgrid.static_grid.hash_table), fg.num_levels, fg.base_res, fg.max_res,
                                    fg.log2_hashmap_size), STATIC_SCALE,
                       [h(l.weight) for l in f.mlp_geo.layers], [h(l.bias) for l in f.mlp_geo.layers],
                       [h(l.weight) for l in f.mlp_feature.layers], [h(l.bias) for l in f.mlp_feature.layers],
                       beta=float(f.sdf_to_density.beta), use_sdf=True)
    return props, fp


def c3_parity(device, n_cam=1024, n_lidar=512):
    """Forward outputs of a slice of the c3 workload (one camera patch + 512 lidar rays, the c3 model and grids) in eval mode
    (no jitter) through the fused eval kernels AND through the fused TRAINING nodes, against the C oracle's chain."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import neurad_oracle as O
    import oracle_c
    from neurad_studio_amd.cameras.rays import RayBundle
    from neurad_studio_amd.models.neurad import NeuRADHotPath, NeuRADHotPathConfig

    torch.manual_seed(11)
    m = NeuRADHotPath(NeuRADHotPathConfig(), static_scale=STATIC_SCALE, num_sensors=7, duration=8.0).to(device)
    with torch.no_grad():
        m.field.hashgrid.static_grid.hash_table.mul_(1000.0)
        for p in m.proposal_fields:
            p.hashgrid.static_grid.hash_table.mul_(2000.0)
    o, d, area, times, md = joint_batch(device, 0, n_cam, n_lidar)
    R = n_cam + n_lidar

    def rb():
        return RayBundle(origins=o, directions=d, pixel_area=area.clone(), nears=torch.zeros((R, 1), device=device), fars=None,
                         times=times, metadata=dict(md))

    m.eval()
    with torch.no_grad():
        ev = m.get_nff_outputs(rb())
    m.train()
    m.sampler.eval()  # the training nodes, without jitter
    tr = m.get_nff_outputs(rb(), calc_lidar_losses=True)
    h = lambda t: t.detach().cpu().numpy()  # noqa: E731
    props, fp = _oracle_params(m, O)
    a = torch.where(md["is_lidar"], area, area * 9.0)
    so = oracle_c.proposal_sampler(props, h(o), h(d), h(a).reshape(-1), np.zeros(R, np.float32), np.full(R, 20000.0, np.float32))
    ref = oracle_c.render_fwd(fp, h(o), h(d), h(a).reshape(-1), so["starts"], so["ends"])
    rl2 = lambda x, y: float(np.linalg.norm(x - y) / np.linalg.norm(y))  # noqa: E731
    return {"eval_features": rl2(h(ev["features"][:, :32]), ref["features"]),
            "eval_accumulation": rl2(h(ev["accumulation"]), ref["accumulation"]),
            "train_nodes_features": rl2(h(tr["features"][:, :32]), ref["features"]),
            "train_nodes_accumulation": rl2(h(tr["accumulation"]), ref["accumulation"]),
            "train_nodes_prop_weights_0": rl2(h(tr["weights_list"][0][..., 0]), so["prop_weights"][0]),
            "tolerance": 1e-4,
            "sample": f"{n_cam} camera + {n_lidar} lidar rays of the c3 batch, NeuRAD-default grids, no jitter: fused eval "
                      "kernels and fused training nodes vs oracle/neurad_oracle_c.c (sampler + field + compositing)",
            "compositing": "unpinned against nerfacc itself (DESIGN.md §3)"}


def c4_parity(m, o, d, area_scaled, times, n=160):
    """Both fused eval kernels of the config[4] scene -- 32 actors, fp16 static + actor tables, THIS model's parameters --
    on the first ``n`` rays of the batch against the numpy oracle's actor path (oracle/neurad_oracle.py: encode_with_actors /
    field_fwd_actors, pinned to the reference's goldens field_actors / proposal_actors): the first proposal round's
    weights (static density with the in-box samples' actor densities spliced in) and, at the sampler's final samples,
    features / depth / accumulation."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import neurad_oracle as O

    h = lambda t: t.detach().float().cpu().numpy()  # noqa: E731
    sky = m.config.sampling.sky_distance
    o, d, a, t = o[:n].contiguous(), d[:n].contiguous(), area_scaled[:n].reshape(-1).contiguous(), times[:n].reshape(-1).contiguous()
    from neurad_studio_amd.cameras.rays import RayBundle

    with torch.no_grad():
        z = torch.zeros(n, device=o.device)
        _, cand = m.field.hashgrid.prepare_actors(o, d, a, torch.stack([z, z + 1], -1), torch.stack([z + 1, z + 2], -1), t)
        rb = RayBundle(origins=o, directions=d, pixel_area=a[:, None], times=t[:, None], nears=z[:, None].clone(),
                       fars=torch.full((n, 1), sky, device=o.device), metadata={})
        pf = m.proposal_fields[-1]
        rs, pw, prs = m.sampler.generate_fused(rb, [pf, pf], sky, actor_cand=cand)
        st = rs.frustums.starts[..., 0].contiguous()
        en = rs.frustums.ends[..., 0].clone()
        en[:, -1] = sky
        feats, depth, acc = m.field.render(o, d, a, st, en, times=t, actor_cand=cand)[:3]
    act = m.field.hashgrid.actors

    def actor_params(hg):
        c = hg.config.actor
        return O.ActorParams(h(act.unique_timestamps), h(act.actor_positions), h(act.actor_rotations_6d),
                             act.actor_present_at_time.cpu().numpy().astype(bool), h(act.actor_sizes), h(act.actor_padding),
                             [O.GridParams(h(g.hash_table), c.num_levels, c.base_res, c.max_res, c.log2_hashmap_size)
                              for g in hg.actor_grids], actor_scale=c.actor_scale)

    props, fp = _oracle_params(m, O)
    rl2 = lambda x, y: float(np.linalg.norm(x - y) / np.linalg.norm(y))  # noqa: E731
    on, dn, an, tn = h(o), h(d), h(a), h(t)
    s0, e0 = h(prs[0].frustums.starts[..., 0]), h(prs[0].frustums.ends[..., 0])
    enc, _ = O.encode_with_actors(props[-1].grid, STATIC_SCALE, actor_params(pf.hashgrid), on, dn, an, s0, e0, tn)
    dens = np.exp(enc @ props[-1].decoder_w.T).reshape(s0.shape).astype(np.float32)
    w0 = O.weights_from_density(e0 - s0, dens)
    ref = O.field_fwd_actors(fp, actor_params(m.field.hashgrid), on, dn, an, h(st), h(en), tn)
    w, _ = O.render_weight_from_alpha(ref["alpha"])
    rf, rd, ra = O.composite(w, ref["feature"], h(st), h(en))
    return {"sampler_round0_weights": rl2(h(pw[0][..., 0]), w0), "eval_features": rl2(h(feats), rf),
            "eval_depth": rl2(h(depth), rd), "eval_accumulation": rl2(h(acc), ra), "tolerance": 1e-4,
            "sample": f"the first {n} rays of the batch, {len(m.field.hashgrid.actor_grids)} actors, fp16 static + actor tables: "
                      "nrhip_proposal_sampler_fwd_actors and nrhip_render_fwd_actors vs oracle/neurad_oracle.py (actor path)",
            "compositing": "unpinned against nerfacc itself (DESIGN.md §3)"}


def actor_scene(n_actors, seed=21):
    """n parked / slowly moving boxes (2 x 4.6 x 1.6 m) on a 60 m square, 9 poses over 4 s each"""
    ts = torch.linspace(0.0, 4.0, 9)
    gen = torch.Generator().manual_seed(seed)
    trajs = []
    for _ in range(n_actors):
        x0, y0 = 60 * torch.rand(2, generator=gen) - 30
        yaw, v = 6.28 * float(torch.rand(1, generator=gen)), 4 * float(torch.rand(1, generator=gen))
        poses = torch.eye(4).repeat(len(ts), 1, 1)
        c, sn = np.cos(yaw), np.sin(yaw)
        poses[:, :3, :3] = torch.tensor([[c, -sn, 0.0], [sn, c, 0.0], [0.0, 0.0, 1.0]])
        poses[:, 0, 3], poses[:, 1, 3], poses[:, 2, 3] = x0 + v * ts * c, y0 + v * ts * sn, 0.8
        trajs.append({"timestamps": ts.clone(), "poses": poses, "dims": torch.tensor([2.0, 4.6, 1.6]),
                      "symmetric": torch.tensor(True), "deformable": torch.tensor(False)})
    return trajs, gen


def bench_c4(args, device, rank, world):
    """BASELINE config[4]: 65 536 rays per GPU, appearance embedding (16-d), 32 dynamic actors, the main field's static and
    actor tables stored as fp16.  Step = one eval pass of get_outputs_for_ray_bundle (fused proposal sampler with per-sample
    actor select + fused field / compositing with per-sample table select + appearance); the line also carries one
    training step of the same scene on all 65 536 rays (fused nodes with row overrides, fp16-storage static + actor tables
    on HashGridAdam's fp32 master copies)."""
    from neurad_studio_amd import ops
    from neurad_studio_amd.cameras.rays import RayBundle
    from neurad_studio_amd.model_components.dynamic_actors import DynamicActors, DynamicActorsConfig
    from neurad_studio_amd.model_components.losses import distortion_loss, zipnerf_interlevel_loss
    from neurad_studio_amd.models.neurad import NeuRADHotPath, NeuRADHotPathConfig

    R, A = 65536, 32
    trajs, gen = actor_scene(A)
    torch.manual_seed(2)
    m = NeuRADHotPath(NeuRADHotPathConfig(), static_scale=STATIC_SCALE, num_sensors=6, duration=4.0,
                      actors=DynamicActors(DynamicActorsConfig(), trajectories=trajs)).to(device).eval()
    with torch.no_grad():
        m.field.hashgrid.static_grid.hash_table.mul_(300.0)
        for gr in m.field.hashgrid.actor_grids:
            gr.hash_table.mul_(2000.0)
        for p in m.proposal_fields:
            p.hashgrid.static_grid.hash_table.mul_(500.0)
        for gr in [m.field.hashgrid.static_grid, *m.field.hashgrid.actor_grids]:
            gr.hash_table.data = gr.hash_table.data.half()
        # round 5: the proposal fields' STATIC tables in fp16 storage as well (their small actor grids stay fp32).  On this
        # incoherent batch the fused sampler runs against the L2 <-> fabric bandwidth (fp32 tables: 15.7 GB per launch,
        # profiles/r05_pmc_traffic.txt), and half the footprint is half the misses.  NRHIP_C4_PROP_FP32=1: the round-4 setup.
        if os.environ.get("NRHIP_C4_PROP_FP32", "0") != "1":
            for p in m.proposal_fields:
                p.hashgrid.static_grid.hash_table.data = p.hashgrid.static_grid.hash_table.data.half()
    # NRHIP_C4_ORDER_RAYS=1: the render stage walks the (incoherent) batch in the order of ops.ray_order.  Measured round 5:
    # traffic of the ACT slice 1.73 -> 1.59 GB per launch, but the stage gets slower, 0.61 -> 0.69 ms -- also with the
    # multi-workgroup ordering pass (profiles/r05_ab.txt, r05_ab_c4_order_large.txt): off
    m.order_rays = os.environ.get("NRHIP_C4_ORDER_RAYS", "0") == "1"
    gen.manual_seed(31 + rank)
    o = (torch.randn(R, 3, generator=gen) * torch.tensor([20.0, 20.0, 0.3]) + torch.tensor([0.0, 0.0, 1.5])).to(device)
    d = torch.nn.functional.normalize(torch.randn(R, 3, generator=gen) * torch.tensor([1.0, 1.0, 0.1]), dim=-1).to(device)
    times = (4 * torch.rand(R, 1, generator=gen)).to(device)
    sens = torch.randint(0, 6, (R, 1), generator=gen).to(device)
    area = torch.full((R, 1), 2.7e-7, device=device)

    def bundle():
        return RayBundle(origins=o, directions=d, pixel_area=area, times=times, metadata={"sensor_idxs": sens})

    state = {}
    ev_every = 4
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(2)] for _ in range((args.steps + ev_every - 1) // ev_every)]
    ev_s = [[torch.cuda.Event(enable_timing=True) for _ in range(2)] for _ in range(len(ev))]
    real_render, real_sampler = ops.render_fwd_actors, ops.proposal_sampler_fwd

    def timed_render(*a, **k):  # HIP events around the render stage (partition + static slice + ACT slice) of timed steps
        e = state.get("ev")
        if e is not None:
            e[0].record()
        r = real_render(*a, **k)
        if e is not None:
            e[1].record()
        return r

    def timed_sampler(*a, **k):  # ... and around the fused proposal sampler (per-sample actor select): the step's largest kernel
        e = state.get("ev_s")
        if e is not None:
            e[0].record()
        r = real_sampler(*a, **k)
        if e is not None:
            e[1].record()
        return r

    def step(i=None):
        state["ev"] = ev[i // ev_every] if i is not None and i % ev_every == 0 else None
        state["ev_s"] = ev_s[i // ev_every] if i is not None and i % ev_every == 0 else None
        state["out"] = m.get_outputs_for_ray_bundle(bundle(), num_rays_per_chunk=1 << 17)

    ops.render_fwd_actors, ops.proposal_sampler_fwd = timed_render, timed_sampler
    try:
        elapsed = timed(step, args.steps, args.warmup, world, device)
    finally:
        ops.render_fwd_actors, ops.proposal_sampler_fwd = real_render, real_sampler
    assert torch.isfinite(state["out"]["features"]).all()
    S = m.config.sampling.num_nerf_samples
    n_field = R * S
    # algorithmic bytes of the render stage: 512 B of fp16 table reads per sample (8 levels x 8 corners x 4 features x 2 B;
    # a sample inside a box reads its actor's 4-level grid instead: 256 B) + 8 B interval + per-ray I/O
    with torch.no_grad():
        rb = bundle()
        m._scale_pixel_area(rb)
        n0 = torch.zeros(R, device=device)
        spec, cand = m.field.hashgrid.prepare_actors(o, d, rb.pixel_area.reshape(-1), torch.stack([n0, n0 + 1], -1),
                                                     torch.stack([n0 + 1, n0 + 2], -1), times.reshape(-1))
        rb.fars = torch.full_like(rb.pixel_area, m.config.sampling.sky_distance)
        rb.nears = torch.zeros_like(rb.fars)
        rs, _, _ = m.sampler.generate_fused(rb, [m.proposal_fields[-1]] * 2, m.config.sampling.sky_distance, actor_cand=cand)
        st_, en_ = rs.frustums.starts[..., 0].contiguous(), rs.frustums.ends[..., 0].contiguous()
        hits = ops.actor_hits(spec, cand, o, d, rb.pixel_area.reshape(-1), st_, en_)
        frac_hit = float((hits[:, 0] >= 0).float().mean())
        rays_with_cand = float((cand[0] > 0).float().mean())
        parity = c4_parity(m, o, d, rb.pixel_area, times) if rank == 0 else None
    per_sample = algorithmic_bytes_per_sample(8, 4, 2, S)
    alg = n_field * (per_sample - frac_hit * 256.0)
    out = None
    if rank == 0:
        k_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
        s_ms = float(np.mean([a.elapsed_time(b) for a, b in ev_s]))
        sc = m.config.sampling
        n_prop = R * sum(sc.num_proposal_samples)
        gathers = n_prop * 6 * 8  # the static proposal grid's 48 four-byte gathers per evaluation (actor lookups not counted)
        out = {
            "metric": "ray-samples/sec (65536 rays, eval, actors + appearance, fp16 tables)",
            "value": world * n_field * args.steps / elapsed, "unit": "ray-samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32 arithmetic, fp16 table storage", "data": "synthetic",
            "config": {"workload": "BASELINE config[4]: 65536 rays per GPU, NeuRAD-default grids with the main field's static + "
                                   f"{A} actor tables in fp16, appearance embedding, {A} dynamic actors, fused proposal sampler "
                                   "(2 rounds, per-sample actor select) + fused field + compositing, eval",
                       "rays_per_gpu": R, "actors": A, "samples_in_a_box": frac_hit, "rays_with_candidates": rays_with_cand,
                       "parallelism": f"rays sharded x{world}, no collective"},
            "rays_per_sec": world * R * args.steps / elapsed,
            # the step's dominant kernel: the fused proposal sampler with the per-sample actor select.  Like config[2]'s it
            # runs against the vector L1's access rate, not HBM (its tables are cache resident)
            "roofline": {"kernel": "nrhip::proposal_sampler_kernel<ACT> (both rounds on chip, one wave per ray, per-sample actor "
                                   "select; HIP events around nrhip_proposal_sampler_fwd_actors in the timed steps)",
                         "bound": "l1", "achieved": gathers / (s_ms * 1e-3) / 1e9, "peak": L1_ACCESS_PEAK_G,
                         "unit": "Gaccess/s", "frac": gathers / (s_ms * 1e-3) / 1e9 / L1_ACCESS_PEAK_G,
                         "traffic": recorded_traffic("proposal_sampler_actors"),
                         "kernel_ms": s_ms, "accesses_per_launch": gathers,
                         "fabric_rate_gb_s": (recorded_traffic("proposal_sampler_actors") or 0) / (s_ms * 1e-3) / 1e9 or None,
                         "fabric_note": "on this incoherent batch (random origins / directions) the proposal table's lines miss the "
                                        "per-XCD L2s: `traffic` / kernel time is the L2 <-> fabric (MALL / HBM) rate the kernel "
                                        "actually runs against (round-5 PMC passes: 15.7 GB per launch at 7.3 TB/s with fp32 "
                                        "tables) -- the L1 figure below is what the same kernel reaches on camera patches (c2)",
                         "what": "algorithmic 4-byte gathers of the STATIC proposal grid (48 per proposal evaluation, "
                                 f"{n_prop} evaluations per launch) per second against the vector L1's access rate; config[2]'s "
                                 "sampler without actors reaches 0.8 of it"},
            "render_roofline": {"kernel": "nrhip_render_fwd_actors: actor_partition + render_kernel<8,4,32,fp16,composite> over the rays "
                                   "without candidates + render_kernel<8,4,32,fp16,composite,ACT> over the rays with",
                         "bound": "hbm", "achieved": alg / (k_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": alg / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": recorded_traffic("render_actors_fp16"),
                         "algorithmic_bytes_per_launch": alg, "kernel_ms": k_ms,
                         "bytes_per_sample": "8 levels x 8 corners x 4 features x 2 B = 512 B (256 B for a sample inside an "
                                             "actor box: its 4-level grid) + 8 B interval + per-ray I/O / S"}}
        out["parity_rel_l2_vs_oracle"] = parity
        out["train_step"] = None
    # ---- the training step of the same scene, whole batch (fused nodes with per-sample row overrides for the in-box samples) ----
    del state["out"]
    torch.cuda.empty_cache()
    m.train()
    params = [p for p in m.parameters() if p.requires_grad]
    opt, opt_name = make_optimizer(params, groups=m.get_param_groups())
    from neurad_studio_amd.parallel.data_parallel import GradientSynchronizer

    sync = GradientSynchronizer(params, average=True, usage="dynamic")
    Rt = int(os.environ.get("NRHIP_C4_TRAIN_RAYS", R))  # the whole batch
    target = torch.rand((Rt, 48), device=device)
    ev_t = [[torch.cuda.Event(enable_timing=True) for _ in range(2)] for _ in range(16)]
    real_field_train = ops.field_fwd_train

    def timed_field_train(*a, **k):  # HIP events around the step's largest kernel (fused training forward of the field)
        e = state.get("ev_t")
        if e is not None:
            e[0].record()
        r = real_field_train(*a, **k)
        if e is not None:
            e[1].record()
        return r

    def tstep(i=None):
        state["ev_t"] = ev_t[i] if i is not None and i < len(ev_t) else None
        rb = RayBundle(origins=o[:Rt], directions=d[:Rt], pixel_area=area[:Rt].clone(), times=times[:Rt],
                       metadata={"sensor_idxs": sens[:Rt]})
        nff = m.get_nff_outputs(rb)
        loss = (5.0 * torch.nn.functional.mse_loss(nff["features"], target)
                + 0.001 * zipnerf_interlevel_loss(nff["weights_list"], nff["ray_samples_list"])
                + 0.002 * distortion_loss(nff["weights_list"], nff["ray_samples_list"]))
        opt.zero_grad(set_to_none=True)
        loss.backward()
        sync.sync()
        opt.step()
        state["loss"] = loss

    tsteps = max(3, min(args.train_steps // 6, 10))
    ops.field_fwd_train = timed_field_train
    try:
        el = timed(tstep, tsteps, 3, world, device)
    finally:
        ops.field_fwd_train = real_field_train
    if os.environ.get("NRHIP_BENCH_TORCH_PROFILE") and rank == 0:
        torch_op_attribution(tstep, os.environ["NRHIP_BENCH_TORCH_PROFILE"])
    if rank == 0:
        f_ms = float(np.mean([a.elapsed_time(b) for a, b in ev_t[:tsteps]]))
        # the fused training forward moves, per field sample: fp16 table reads (8 levels x 8 corners x 4 features x 2 B; the
        # in-box samples take their rows from the override list instead) + 8 B interval + saved activations + outputs
        per = 8 * 8 * 4 * 2 + 8 + (32 + 32 + 48 + 64) * 4 + 34 * 4
        t_alg = Rt * S * per
        out["train_step"] = {"ms_per_iter": el / tsteps * 1e3, "iters_per_sec": tsteps / el, "rays_per_gpu": Rt,
                             "us_per_ray": el / tsteps * 1e6 / Rt,
                             "rays_per_sec": world * Rt * tsteps / el, "optimizer": opt_name, "loss_finite": bool(torch.isfinite(state["loss"])),
                             "roofline": {"kernel": "nrhip::render_kernel<8,4,32,fp16,train,OVR> (fused training forward of the field with "
                                                    "row overrides; HIP events around nrhip_field_fwd_train_ovr in the timed steps)",
                                          "bound": "hbm", "achieved": t_alg / (f_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                          "frac": t_alg / (f_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                          "traffic": recorded_traffic("field_fwd_train_ovr_fp16"), "kernel_ms": f_ms,
                                          "algorithmic_bytes_per_launch": t_alg},
                             "what": f"all {Rt} rays of the scene in one step: sampler rounds with the actor overlay (in-box "
                                     "samples spliced in by nrhip_actor_density_splice_*), fused field + SDF head + compositing + "
                                     "appearance with per-sample row overrides for the in-box samples, feature / interlevel / "
                                     "distortion losses, backward (actor grids, trajectories), the reference's optimizer groups; the "
                                     f"main field's static table and its {A} actor grids are fp16 storage (HashGridAdam: fp32 master "
                                     "copies in the optimizer state, the fp16 gradient read and the fp16 table written inside the "
                                     "kernel); the proposal fields' static tables are fp16 storage too (NRHIP_C4_PROP_FP32=1: fp32), their actor grids fp32"}
    return out


def self_launch(n: int) -> int:
    """re-run this command line as n ranks of one node (torch.distributed.run, rendezvous on 127.0.0.1 at a free port) and
    return the launcher's exit code"""
    import socket
    import subprocess

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", "1")  # (what torch.distributed.run would set itself, with a warning)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: the only form this host driver supports
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", choices=["c1", "c2", "c3", "c4"], default="c1")
    ap.add_argument("--sharded-adam", action="store_true",
                    help="N > 1: hash tables on ShardedTableAdam (reduce-scatter of the gradient, Adam on 1/N of each table, "
                         "all-gather of the parameters) instead of gradient all-reduce + a full-table Adam on every rank")
    ap.add_argument("--wire-bf16", action="store_true",
                    help="N > 1: the reduce-scatter leg of the large table gradients in bf16 (one rounding per rank, all-to-all to "
                         "the owning rank, fp32 sum there; fp32 all-gather of the mean gradient, or with --sharded-adam of the "
                         "updated parameters): 6 instead of 8 bytes per element per step, replicas bit-identical")
    ap.add_argument("--sparse-exchange", action="store_true",
                    help="N > 1: coarse hash-table levels travel as (row, values) lists (GradientSynchronizer level_tables)")
    ap.add_argument("--torch-decoder", action="store_true",
                    help="train_full / c3: the RGB decoder on the torch modules (MIOpen, fp16 autocast) instead of the HIP kernels")
    ap.add_argument("--no-rgb-decoder", action="store_true", help="train_full / c3 without the RGB CNN decoder (round-2 step)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true",
                    help="train_full / c3 on one GPU: time the eagerly launched step only (default: the step captured in a HIP "
                         "graph, the eager figure reported beside it)")
    ap.add_argument("--no-train", action="store_true", help="skip the train iters/sec sections")
    ap.add_argument("--no-variants", action="store_true",
                    help="c1: skip the two labelled non-headline launches of the render kernel (profiling runs: their "
                         "launches would enter the per-kernel averages)")
    ap.add_argument("--via-plugin", action="store_true",
                    help="c3: ALSO time the step as `ns-train neurad-hip` executes it -- the reference's Trainer.train_iteration "
                         "(autocast + GradScaler + Optimizers) over the plugin model (needs nerfstudio importable)")
    ap.add_argument("--via-plugin-only", action="store_true", help="c3: time only the --via-plugin step (kernel traces)")
    ap.add_argument("--train-steps", type=int, default=60)
    ap.add_argument("--train-full-steps", type=int, default=30, help="0 skips the train_full section")
    args = ap.parse_args()
    global WIRE_DTYPE
    WIRE_DTYPE = torch.bfloat16 if args.wire_bf16 else None

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` as a PLAIN command: spawn the N ranks here, as the reference's own entry point does
        # (scripts/train.py:167-230 launches its workers itself) -- one process per GPU under torch.distributed.run on a free
        # local port; rank 0 of the children prints the one JSON line on this process's stdout.  The explicit
        # `python -m torch.distributed.run ... bench.py --gpus N` form sets WORLD_SIZE and never reaches this branch.
        sys.exit(self_launch(args.gpus))
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world} (launch with torch.distributed.run)"
    # NRHIP_DIST_BACKEND=gloo: REHEARSAL of the N > 1 code path on a box with fewer GPUs than ranks (all ranks share the
    # visible devices, collectives run over gloo on CUDA tensors) -- it exercises init_process_group, the gradient hooks on
    # the real autograd nodes, the sharded optimizer and the max-over-ranks timing; it is NOT a scaling measurement.
    backend = os.environ.get("NRHIP_DIST_BACKEND", "nccl")
    n_dev = max(torch.cuda.device_count(), 1)
    rehearsal = world > 1 and (backend != "nccl" or n_dev < world)
    dev_index = local_rank % n_dev
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    dist_info = None
    if world > 1:
        import socket

        import torch.distributed as dist

        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)  # "nccl" == RCCL on ROCm
        else:
            dist.init_process_group(backend)
        # fail loudly on a mis-launch instead of timing N replicas of one GPU: the group has --gpus ranks and (outside the
        # gloo rehearsal) every rank of a host owns a different device
        if dist.get_world_size() != args.gpus:
            raise RuntimeError(f"process group has {dist.get_world_size()} ranks, --gpus {args.gpus}")
        try:
            uuid = str(torch.cuda.get_device_properties(dev_index).uuid)
        except Exception:  # noqa: BLE001
            uuid = f"index{dev_index}"
        mine = (socket.gethostname(), torch.cuda.current_device(), uuid)
        everyone = [None] * world
        dist.all_gather_object(everyone, mine)
        if not rehearsal and len({(h, u) for h, _, u in everyone}) != world:
            raise RuntimeError(f"ranks share a GPU: {everyone} (launch one rank per device, or set NRHIP_DIST_BACKEND=gloo for "
                               "the labelled one-GPU rehearsal)")
        try:
            rccl = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:  # noqa: BLE001
            rccl = None
        dist_info = {"backend": backend, "rccl_version": rccl, "devices": [f"{h}:{i}" for h, i, _ in everyone],
                     "env": {k: v for k, v in os.environ.items() if k.startswith(("NCCL_", "RCCL_", "HSA_", "HIP_VISIBLE", "ROCR_VISIBLE"))}}
    dev_state = device_state(dev_index) if rank == 0 else None

    if args.config == "c2":
        out = bench_c2(args, device, rank, world)
    elif args.config == "c4":
        out = bench_c4(args, device, rank, world)
    elif args.config == "c3" and args.via_plugin_only:  # (profiling runs: only the plugin's step in the kernel trace)
        steps = min(args.steps, 50)
        vp = train_via_plugin_section(device, rank, world, steps, max(2, min(args.warmup, 5)))
        out = {"metric": "train iters/sec (camera+lidar joint batch), the step as `ns-train neurad-hip` executes it",
               "value": vp.get("rays_per_sec"), "unit": "rays/s", "n_gpus": world, "steps": steps,
               "warmup": max(2, min(args.warmup, 5)), "ms_per_step": vp.get("ms_per_iter"), "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": vp.get("what")}, "train_via_plugin": vp}
    elif args.config == "c3":
        steps = min(args.steps, 50)
        tf = train_full_section(device, rank, world, steps, max(2, min(args.warmup, 5)), rgb_decoder=not args.no_rgb_decoder,
                                sharded_adam=args.sharded_adam, sparse_exchange=args.sparse_exchange,
                                torch_decoder=args.torch_decoder, graph=not args.no_graph)
        out = {"metric": "train iters/sec (camera+lidar joint batch)", "value": tf["rays_per_sec"], "unit": "rays/s",
               "n_gpus": world, "steps": steps, "warmup": max(2, min(args.warmup, 5)), "ms_per_step": tf["ms_per_iter"],
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": tf["what"], "rays_per_gpu": tf["rays_per_gpu"],
                          "parallelism": f"dp{world}: rays sharded, table gradients reduce-scatter + all-gather"},
               "iters_per_sec": tf["iters_per_sec"], "roofline": tf.pop("roofline"), "train_full": tf}
        if args.via_plugin:
            import gc

            gc.collect()
            torch.cuda.empty_cache()
            vp = train_via_plugin_section(device, rank, world, steps, max(2, min(args.warmup, 5)))
            if "ms_per_iter" in vp:
                vp["ratio_to_train_full"] = vp["ms_per_iter"] / tf["ms_per_iter"]
            out["train_via_plugin"] = vp
        if rank == 0 and world == 1:
            out["parity_rel_l2_vs_oracle"] = c3_parity(device)
    else:
        out, (fs, origins, dirs, area, edges, feats) = bench_c1(args, device, rank, world)
        train = train_full = None
        def guarded(section, *a):
            """one GPU: errors propagate.  N > 1 (a launch this container cannot rehearse): an exception in an auxiliary
            training section is reported IN the line instead of costing the headline measurement"""
            if world == 1:
                return section(*a)
            try:
                return section(*a)
            except Exception as e:  # noqa: BLE001
                return {"error": f"{type(e).__name__}: {e}"[:500]}

        if not args.no_train:
            train = guarded(train_section, device, rank, world, args.train_steps, max(5, args.warmup // 2))
            if isinstance(train, dict) and "error" not in train:
                soft = guarded(lambda: train_section(device, rank, world, max(args.train_steps // 2, 10), 5, beta=3.0,
                                                     table_scale=1.0))
                train["non_saturating"] = dict({k: soft[k] for k in ("iters_per_sec", "ms_per_iter", "error") if k in soft},
                                               what="the same step on a scene that does not saturate: beta = 3 and an O(1) "
                                                    "table (the default section's beta = 20 on random weights turns every "
                                                    "ray opaque; 76 % of its samples then carry an exactly-zero gradient "
                                                    "and are dropped by the table-gradient partition, DESIGN §5)")
            if args.train_full_steps > 0:
                # the previous sections' buffers go back to the driver first: with them cached, the allocator was seen to
                # fall back to fresh hipMallocs inside the first timed steps on a fresh box (19 instead of 11.6 ms/iter)
                import gc

                gc.collect()
                torch.cuda.empty_cache()
                train_full = guarded(lambda: train_full_section(device, rank, world, args.train_full_steps, 8,
                                                                rgb_decoder=not args.no_rgb_decoder,
                                                                sharded_adam=args.sharded_adam, sparse_exchange=args.sparse_exchange,
                                                                torch_decoder=args.torch_decoder, graph=not args.no_graph))
                if not args.no_rgb_decoder and isinstance(train_full, dict) and "error" not in train_full:
                    gc.collect()
                    torch.cuda.empty_cache()
                    hot = guarded(lambda: train_full_section(device, rank, world, max(args.train_full_steps // 2, 5), 5,
                                                             rgb_decoder=False, sharded_adam=args.sharded_adam, sparse_exchange=args.sparse_exchange))
                    if isinstance(hot, dict):
                        hot.pop("roofline", None)
                        train_full["hot_path_only"] = {k: hot[k] for k in ("iters_per_sec", "ms_per_iter", "rays_per_sec", "error")
                                                       if k in hot}
                        train_full["hot_path_only"]["what"] = ("the same step without the RGB CNN decoder (a feature "
                                                               "regression stands in for it and the rgb loss): round 2's step")
        if rank == 0:
            if train is not None:
                out["train"] = train
            if train_full is not None:
                if "roofline" in train_full:
                    train_full["field_forward_roofline"] = train_full.pop("roofline")
                out["train_full"] = train_full
            if world == 1 and not args.no_cpu_baseline:
                cb, (n, ref) = cpu_baseline(fs, origins, dirs, area, edges)
                rt = reference_torch_cpu()
                if rt is not None and rt.get("where", "").startswith("this host"):
                    # north_star's CPU leg: the reference's own torch field eval on this box's cores; the C/OpenMP port of
                    # the oracle (the checker of this line's parity figure) is reported beside it
                    out["cpu_baseline"] = rt
                    out["cpu_port_c"] = cb
                else:
                    out["cpu_baseline"] = cb
                    if rt is not None:
                        out["reference_torch_cpu"] = rt
                    # no reference tree on this box: its torch formulation, restated, timed here on the host cores
                    try:
                        out["reference_torch_cpu_port_here"] = torch_port_cpu(fs, origins, dirs, area, edges)
                    except Exception as e:  # noqa: BLE001  (a baseline leg must not cost the headline line)
                        out["reference_torch_cpu_port_here"] = {"error": f"{type(e).__name__}: {e}"[:300]}
                err = float(np.linalg.norm(feats[:n].cpu().numpy() - ref["features"]) / np.linalg.norm(ref["features"]))
                out["parity_rel_l2_vs_oracle"] = {
                    "features": err, "tolerance": 1e-4,
                    "compositing": "unpinned: nerfacc 0.5.2 is not vendored and the reference substitutes a constant on "
                                   "CPU (models/neurad.py:713-715); the oracle restates its dense formulas (DESIGN.md §3)",
                    "field": "oracle pinned to the reference's own outputs (tests/golden/, oracle/make_golden*.py)"}
    if rank == 0:
        out["device_state"] = dev_state
        if dist_info is not None:
            out["dist"] = dist_info
        if rehearsal:
            out["rehearsal"] = (f"{world} ranks over {backend} on {n_dev} visible GPU(s): a run of the N > 1 code path (process "
                                "group, gradient hooks, exchange, max-over-ranks timing), NOT a scaling measurement")
        if TRAFFIC_NOTES:
            out["traffic_notes"] = TRAFFIC_NOTES
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist

        dist.destroy_process_group()


if __name__ == "__main__":
    main()
