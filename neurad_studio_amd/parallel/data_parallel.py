"""Data parallelism of the hot path: rays shard, parameters replicate, ONE exchange per training step --
the gradient all-reduce (reference: DDP over NCCL, pipelines/base_pipeline.py:304-307).

MI355X-first choices (xGMI is point-to-point, 7 links x ~153 GB/s per GPU; a ring is per-link bound):
  * the hash-table gradients (>= 99.9 % of the bytes: 537 MB fp32 for the default static grid) are reduced as
    FEW, LARGE flat buffers -- `reduce_scatter_tensor` + `all_gather_into_tensor` on a pre-allocated flat buffer, so
    RCCL can drive all links concurrently -- instead of DDP's 25 MB buckets;
  * the < 1 MB of MLP / decoder / embedding gradients travel as one coalesced all-reduce;
  * parameters that got no gradient this step (the never-evaluated proposal_fields[0], models/neurad.py:248)
    are skipped symmetrically on every rank -- what DDP's find_unused_parameters=True does with a bitmap.
Backend-agnostic (``"nccl"`` == RCCL on ROCm, ``"gloo"`` in the CPU tests)."""
from __future__ import annotations

from typing import Iterable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist
from torch import Tensor


def shard_range(n_items: int, rank: int, world: int, granule: int = 1) -> Tuple[int, int]:
    """Contiguous [start, end) slice of ``n_items`` for ``rank``; boundaries are multiples of ``granule`` so
    that e.g. 32x32 camera patches stay whole (neurad.py:362-365)."""
    n_gran = (n_items + granule - 1) // granule
    base, rem = divmod(n_gran, world)
    start = rank * base + min(rank, rem)
    end = start + base + (1 if rank < rem else 0)
    return min(start * granule, n_items), min(end * granule, n_items)


class GradientSynchronizer:
    """Averages (or sums) ``param.grad`` across ranks with large flat collectives."""

    def __init__(self, params: Iterable[torch.nn.Parameter], process_group=None, average: bool = True,
                 large_threshold_bytes: int = 8 << 20) -> None:
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        self.group = process_group
        self.average = average
        self.large_threshold = large_threshold_bytes
        self._flat: Optional[Tensor] = None

    def world_size(self) -> int:
        return dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1

    def _used_mask(self) -> List[bool]:
        """Agree on which parameters have a gradient on ANY rank (missing ones are treated as zeros)."""
        dev = self.params[0].device
        m = torch.tensor([0 if p.grad is None else 1 for p in self.params], device=dev, dtype=torch.int32)
        dist.all_reduce(m, op=dist.ReduceOp.MAX, group=self.group)
        return [bool(v) for v in m.tolist()]

    @torch.no_grad()
    def sync(self) -> int:
        """All-reduce every gradient; returns the number of payload bytes exchanged per rank."""
        world = self.world_size()
        if world == 1 or not self.params:
            return 0
        used = self._used_mask()
        small, nbytes = [], 0
        for p, u in zip(self.params, used):
            if not u:
                continue
            if p.grad is None:
                p.grad = torch.zeros_like(p)
            g = p.grad
            nbytes += g.numel() * g.element_size()
            if g.numel() * g.element_size() >= self.large_threshold and g.is_contiguous() and g.numel() % world == 0:
                # reduce-scatter + all-gather in place on the gradient's own storage: every link busy, no staging copy
                flat = g.view(-1)
                shard = flat.new_empty(flat.numel() // world)
                dist.reduce_scatter_tensor(shard, flat, op=dist.ReduceOp.SUM, group=self.group)
                if self.average:
                    shard.div_(world)
                dist.all_gather_into_tensor(flat, shard, group=self.group)
            else:
                small.append(g)
        if small:
            flat = torch.cat([g.reshape(-1) for g in small])
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
            if self.average:
                flat.div_(world)
            off = 0
            for g in small:
                g.copy_(flat[off:off + g.numel()].view_as(g))
                off += g.numel()
        return nbytes
