"""Sharded optimizer step for the hash tables (opt-in; DESIGN §7 "next step"): reduce-scatter the table GRADIENT, run Adam
on this rank's 1/N shard of the table only, all-gather the updated PARAMETERS.

Against GradientSynchronizer + HashGridAdam (reduce-scatter + all-gather of the gradient, then a full-table Adam on every
rank) the bytes on the wire are the same, but each rank streams 1/N of the 537 MB table and of its two moment arrays through
the optimizer kernel (0.66 ms -> 0.08 ms per step at N = 8 on the NeuRAD-default grid) and keeps 1/N of the moments
(1.07 GB -> 134 MB).  The arithmetic per element is unchanged -- the same ``nrhip_adam_step`` kernel on a slice, the 1/N
gradient average folded into its ``grad_scale`` -- so N ranks end a step with exactly the parameters one process would
have computed from the mean gradient.

``state_dict()`` gathers the moment shards back into full tensors in torch.optim.Adam's layout (the reference's
checkpoints, engine/trainer.py:499-533, engine/optimizers.py:168-181, stay loadable); ``load_state_dict`` scatters them.
Backend-agnostic like data_parallel.py ("nccl" == RCCL; "gloo" in the CPU tests, which inject a torch ``update_fn`` because
the product kernel has no CPU path).  bench.py --sharded-adam uses it for the tables at N > 1."""
from __future__ import annotations

from typing import Callable, Dict, Iterable, List, Optional, Tuple

import torch
import torch.distributed as dist
from torch import Tensor

from .data_parallel import all_gather_flat


def _hip_update(param: Tensor, grad: Tensor, exp_avg: Tensor, exp_avg_sq: Tensor, step: int, lr: float, beta1: float,
                beta2: float, eps: float, weight_decay: float, grad_scale: float) -> None:
    from .. import ops

    ops.adam_step(param, grad, exp_avg, exp_avg_sq, step, lr, beta1, beta2, eps, weight_decay, grad_scale)


class ShardedTableAdam:
    """Adam / AdamW for large fp32 tables whose element count divides by the world size.  Every rank calls ``step()`` after
    backward with its LOCAL ``table.grad``; a table without a gradient on this rank contributes zeros, a table without a
    gradient on ANY rank is skipped -- no moment decay, no step count, like torch.optim.Adam on ``grad is None`` (an actor
    grid no ray hit; all ranks must hold the same list of tables).  usage="dynamic" agrees on that set every step (a tiny
    MAX all-reduce + a host read), "static" once (afterwards only a host-side check that the local pattern is unchanged)."""

    def __init__(self, tables: Iterable[torch.nn.Parameter], lr: float = 1e-2, betas: Tuple[float, float] = (0.9, 0.999),
                 eps: float = 1e-15, weight_decay: float = 0.0, process_group=None, average: bool = True,
                 update_fn: Optional[Callable] = None, usage: str = "dynamic",
                 wire_dtype: Optional[torch.dtype] = None) -> None:
        if usage not in ("dynamic", "static"):
            raise ValueError("usage must be 'dynamic' or 'static'")
        self.usage = usage
        if wire_dtype not in (None, torch.bfloat16):  # (fp16 would overflow under a GradScaler's 2^16: data_parallel.py)
            raise ValueError("wire_dtype: None (fp32) or torch.bfloat16")
        # 16-bit gradient leg (data_parallel.scatter_16bit_*): rounded once per rank, summed in fp32 on the owning rank, whose
        # Adam step then runs on that fp32 sum; the parameters come back in fp32: 6 instead of 8 bytes per element per step
        self.wire_dtype = wire_dtype
        self._agreed: Optional[List[bool]] = None
        self._local_at_agreement: Optional[List[bool]] = None
        self.tables: List[torch.nn.Parameter] = list(tables)
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.group, self.average = process_group, average
        self.update_fn = update_fn if update_fn is not None else _hip_update
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        self.rank = dist.get_rank(process_group) if self.world > 1 else 0
        self.state: List[Dict] = []  # per table: its own step count (tables may be skipped) + the moment shards
        for p in self.tables:
            if p.dtype != torch.float32 or not p.is_contiguous():
                raise ValueError("ShardedTableAdam: contiguous fp32 tables only (fp16-storage tables keep HashGridAdam)")
            if p.numel() % self.world or (p.numel() // self.world) % 4:
                raise ValueError(f"ShardedTableAdam: {p.numel()} elements do not split into {self.world} 16-byte aligned shards")
            m = p.numel() // self.world
            self.state.append({"step": 0, "exp_avg": torch.zeros((m,), device=p.device),
                               "exp_avg_sq": torch.zeros((m,), device=p.device)})

    def _shard(self, flat: Tensor) -> Tensor:
        m = flat.numel() // self.world
        return flat[self.rank * m:(self.rank + 1) * m]

    def _used(self) -> List[bool]:
        local = [p.grad is not None for p in self.tables]
        if self.world == 1:
            return local
        if self.usage == "static" and self._agreed is not None:
            if local != self._local_at_agreement:
                raise RuntimeError("ShardedTableAdam(usage='static'): the set of tables with a gradient changed on this rank")
            return self._agreed
        m = torch.tensor([int(u) for u in local], device=self.tables[0].device, dtype=torch.int32)
        dist.all_reduce(m, op=dist.ReduceOp.MAX, group=self.group)
        used = [bool(v) for v in m.tolist()]
        if self.usage == "static":
            self._agreed, self._local_at_agreement = used, local
        return used

    @torch.no_grad()
    def step(self) -> int:
        """-> payload bytes this rank exchanged (reduce-scatter + all-gather)"""
        nbytes = 0
        b1, b2 = self.betas
        for p, st, used in zip(self.tables, self.state, self._used()):
            if not used:
                continue  # no gradient anywhere: parameters, moments and the step count stay as they are
            st["step"] += 1
            flat_p = p.data.view(-1)
            grad = p.grad if p.grad is not None else torch.zeros_like(p)
            flat_g = grad.contiguous().view(-1)
            if self.world > 1 and self.wire_dtype is not None:
                from .data_parallel import scatter_16bit_finish, scatter_16bit_start

                _, recv = scatter_16bit_start(flat_g, self.world, self.group, False, self.wire_dtype)
                g_shard = scatter_16bit_finish(recv, self.wire_dtype)
                nbytes += flat_g.numel() * (2 + 4) * (self.world - 1) // self.world
            elif self.world > 1:
                from .data_parallel import reduce_scatter_flat

                _, g_shard = reduce_scatter_flat(flat_g, self.world, self.group)
                if g_shard is None:  # no reduce-scatter on this backend / device: flat_g holds the full sum
                    g_shard = self._shard(flat_g)
                nbytes += 2 * flat_g.numel() * 4 * (self.world - 1) // self.world
            else:
                g_shard = flat_g
            p_shard = self._shard(flat_p)  # a view: the update lands in the table itself
            self.update_fn(p_shard, g_shard, st["exp_avg"], st["exp_avg_sq"], st["step"], self.lr, b1, b2, self.eps,
                           self.weight_decay, 1.0 / self.world if (self.average and self.world > 1) else 1.0)
            if self.world > 1:
                # in place (send buffer = this rank's slot of the receive buffer) where the backend supports it
                src = p_shard if dist.get_backend(self.group) == "nccl" else p_shard.clone()
                all_gather_flat(flat_p, src, self.world, self.group)
        return nbytes

    def zero_grad(self, set_to_none: bool = True) -> None:
        for p in self.tables:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    # ---- checkpoints in torch.optim.Adam's layout ------------------------------------------------------------------
    def _gather(self, shard: Tensor, like: Tensor) -> Tensor:
        if self.world == 1:
            return shard.clone().view_as(like)
        full = shard.new_empty(like.numel())
        all_gather_flat(full, shard.contiguous(), self.world, self.group)
        return full.view_as(like)

    def state_dict(self) -> Dict:
        """collective: every rank must call it.  {"state": {i: {"step", "exp_avg", "exp_avg_sq"}}, "param_groups": [...]}
        with FULL-size moments, as torch.optim.Adam over the same tables would write it"""
        state = {i: {"step": torch.tensor(float(st["step"])), "exp_avg": self._gather(st["exp_avg"], p),
                     "exp_avg_sq": self._gather(st["exp_avg_sq"], p)}
                 for i, (p, st) in enumerate(zip(self.tables, self.state))}
        group = {"lr": self.lr, "betas": self.betas, "eps": self.eps, "weight_decay": self.weight_decay, "amsgrad": False,
                 "maximize": False, "params": list(range(len(self.tables)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd: Dict) -> None:
        g = sd["param_groups"][0]
        self.lr, self.betas, self.eps, self.weight_decay = g["lr"], tuple(g["betas"]), g["eps"], g["weight_decay"]
        for i, (p, st) in enumerate(zip(self.tables, self.state)):
            src = sd["state"].get(i)
            if src is None:
                continue
            st["step"] = int(float(src["step"]))
            for k in ("exp_avg", "exp_avg_sq"):
                st[k].copy_(self._shard(src[k].to(p.device, torch.float32).reshape(-1)))
