"""Data parallelism of the hot path: rays shard, parameters replicate, ONE exchange per training step --
the gradient all-reduce (reference: DDP over NCCL, pipelines/base_pipeline.py:304-307).

MI355X-first choices (xGMI is point-to-point, 7 links x ~153 GB/s per GPU; a ring is per-link bound):
  * the hash-table gradients (>= 99.9 % of the bytes: 537 MB fp32 for the default static grid) are reduced as
    FEW, LARGE flat buffers -- `reduce_scatter_tensor` + `all_gather_into_tensor` in place on the gradient's own storage,
    so RCCL can drive all links concurrently -- instead of DDP's 25 MB buckets;
  * OVERLAP: with ``overlap=True`` each large gradient's exchange is enqueued (async) from a post-accumulate-grad hook
    the moment autograd has finished that gradient -- the two 25 MB proposal-table reductions then run under the main
    field's backward instead of after it; ``sync()`` only waits for them and exchanges the rest;
  * the < 1 MB of MLP / decoder / embedding gradients travel as one coalesced all-reduce;
  * parameters without a gradient (the never-evaluated proposal_fields[0], models/neurad.py:248; actor grids no ray
    of the batch hits) are skipped symmetrically on every rank -- what DDP's find_unused_parameters=True does with a
    bitmap.  ``usage="dynamic"`` agrees on that set every step (one tiny MAX all-reduce + a host read: needed when actor
    grids come and go); ``usage="static"`` agrees once and from then on only checks, on the host and without any
    device read, that the local pattern has not changed (static scenes: no per-step host sync at all).
  * OPT-IN, ``level_tables=`` (bench.py --sparse-exchange): a hash table's COARSE levels receive few distinct rows per step
    (config[3] step, scripts/grad_density.py: 0.3 / 1 / 3 / 8 / 17 / 33 / 52 / 67 % of the rows of the field table's eight
    levels), so those levels travel as (row, values) lists in one all-gather and every rank adds the lists in rank order
    (bit-identical replicas); the fine levels stay dense.  Which levels go as lists is decided per step from the agreed
    maximum row count: a list pays while (N-1) * rows * (4 + 4F) bytes < the 2 (N-1)/N * T * 4F of the dense pair (a level the
    hook already started densely finishes densely this step and is re-decided, from this step's counts, for the next).
    On the GPU the counting, the ordered compaction and the merge are csrc/grad_rows.hip (no torch op chains); with
    ``overlap=True`` the table's hook starts the count agreement and the reduce-scatter of the levels that went densely in the
    previous step -- the bulk of the bytes -- under the rest of the backward; sync() only sizes and sends the lists.
Backend-agnostic (``"nccl"`` == RCCL on ROCm, ``"gloo"`` in the CPU tests)."""
from __future__ import annotations

from typing import Dict, Iterable, List, Optional, Tuple

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


def flat_collectives_supported(group, device: torch.device) -> bool:
    """Does the group's backend implement reduce_scatter_tensor / all_gather_into_tensor for tensors on ``device``?
    Decided from (backend, device type) alone -- the same answer on every rank, no exception probing: a rank-local
    RuntimeError (a transport error, a bad shape) must surface, not silently switch ONE rank to a different collective.
    The one unsupported pairing in use is gloo over GPU tensors (the one-GPU rehearsal of the N > 1 path)."""
    return not (dist.get_backend(group) == "gloo" and device.type == "cuda")


def reduce_scatter_flat(flat: Tensor, world: int, group=None, async_op: bool = False):
    """SUM-reduce-scatter of a flat buffer -> (work | None, this rank's shard).  A backend without reduce_scatter_tensor
    for the buffer's device (``flat_collectives_supported``) gets an all-reduce of the whole buffer instead and the shard
    comes back as None: ``flat`` then already holds the full sum."""
    if flat_collectives_supported(group, flat.device):
        shard = flat.new_empty(flat.numel() // world)
        work = dist.reduce_scatter_tensor(shard, flat, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
        return (work if async_op else None), shard
    work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
    return (work if async_op else None), None


def scatter_16bit_start(flat: Tensor, world: int, group=None, async_op: bool = False, wire_dtype=torch.bfloat16):
    """16-bit wire format of the reduce-scatter leg (opt-in; SURVEY §8e prices the table exchange at half the bytes): every
    rank rounds its gradient to ``wire_dtype`` ONCE and sends chunk j straight to its owner j (an all-to-all: on xGMI's
    point-to-point links every pair has its own link, no ring hops, no re-rounding per hop); ``scatter_16bit_finish`` sums
    the N received chunks in fp32, in rank order.  -> (work | None, receive buffer [world, m] as int16 bits).
    The 16-bit payload travels as raw bytes so that any backend's all-to-all carries it (gloo has neither bf16 nor int16)."""
    m = flat.numel() // world
    send = flat.view(world, m).to(wire_dtype).view(torch.int16)
    recv = torch.empty_like(send)
    work = dist.all_to_all_single(recv.view(torch.uint8).view(-1), send.view(torch.uint8).view(-1), group=group,
                                  async_op=async_op)
    return (work if async_op else None), recv


def scatter_16bit_finish(recv: Tensor, wire_dtype=torch.bfloat16) -> Tensor:
    """fp32 sum over the ranks' 16-bit chunks, in rank order (one owner per shard: whatever the order, the all-gather that
    follows hands every replica the same bits) -> this rank's fp32 shard of the SUM"""
    acc = recv[0].view(wire_dtype).float()
    for k in range(1, recv.shape[0]):
        acc += recv[k].view(wire_dtype)
    return acc


def all_gather_flat(full: Tensor, shard: Tensor, world: int, group=None) -> None:
    """all_gather_into_tensor, or -- where the backend lacks it for this device -- a list gather + copy"""
    if flat_collectives_supported(group, full.device):
        dist.all_gather_into_tensor(full, shard, group=group)
        return
    parts = [torch.empty_like(shard) for _ in range(world)]
    dist.all_gather(parts, shard, group=group)
    full.copy_(torch.cat(parts))


class GradientSynchronizer:
    """Averages (or sums) ``param.grad`` across ranks with large flat collectives."""

    def __init__(self, params: Iterable[torch.nn.Parameter], process_group=None, average: bool = True,
                 large_threshold_bytes: int = 8 << 20, usage: str = "dynamic", overlap: bool = False,
                 skip: Iterable[torch.nn.Parameter] = (),
                 level_tables: Optional[Dict[torch.nn.Parameter, int]] = None, profile: bool = False,
                 wire_dtype: Optional[torch.dtype] = None, auto_sync: bool = False) -> None:
        if usage not in ("dynamic", "static"):
            raise ValueError("usage must be 'dynamic' or 'static'")
        if overlap and usage != "static":
            raise ValueError("overlap=True needs usage='static': a hook cannot wait for the other ranks' usage bitmap")
        skipped = {id(p) for p in skip}  # exchanged by someone else (parallel/sharded_adam.py reduces its own tables)
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad and id(p) not in skipped]
        self.group = process_group
        self.average = average
        self.large_threshold = large_threshold_bytes
        self.usage = usage
        self.overlap = overlap
        if wire_dtype not in (None, torch.bfloat16):
            # fp16 is not offered: under the trainer's GradScaler the table gradients carry a 2^16 scale, a single rounding to
            # fp16 overflows to inf on the wire and the step is skipped on every rank
            raise ValueError("wire_dtype: None (fp32) or torch.bfloat16")
        if wire_dtype is not None and level_tables:
            import warnings

            warnings.warn("GradientSynchronizer: wire_dtype applies to plain large gradients only; tables listed in "
                          "level_tables keep fp32 on the wire (their dense level runs go through the fp32 reduce-scatter)",
                          stacklevel=2)
        # 16-bit reduce-scatter leg of the LARGE fp32 gradients (scatter_16bit_*): rounded once per rank, summed in fp32 on
        # the owning rank, the fp32 mean all-gathered -- replicas stay bit-identical, 6 instead of 8 bytes per element on
        # the wire.  Level tables and small gradients keep fp32.
        self.wire_dtype = wire_dtype
        self._agreed: Optional[List[bool]] = None      # usage == "static": the set agreed at the first sync
        self._agreed_all: Optional[List[bool]] = None  # ... and the parameters EVERY rank holds a local gradient for
        self._local_at_agreement: Optional[List[bool]] = None
        self._inflight: Dict[int, tuple] = {}  # param index -> (work, flat grad, shard | 16-bit receive buffer, is 16-bit)
        self._inflight_levels: Dict[int, tuple] = {}   # level table -> (counts, count all-reduce, compaction aux, dense runs)
        self._dense_prev: Dict[int, set] = {}          # level table -> the levels that went densely in the previous step
        self.overlapped_level_runs_last_step = 0       # dense level runs whose exchange a hook started (last sync())
        self._hooks = []
        self.overlapped_last_step = 0
        # level-sparse exchange (opt-in): parameter index -> number of levels of a [levels * T, F] hash table
        self._levels: Dict[int, int] = {}
        for p, n_levels in (level_tables or {}).items():
            i = next((k for k, q in enumerate(self.params) if q is p), None)
            if i is not None:
                if p.dim() != 2 or p.shape[0] % n_levels or not p.is_contiguous():
                    raise ValueError("level_tables: contiguous [levels * T, F] tables only")
                self._levels[i] = int(n_levels)
        self.last_wire_bytes = 0        # bytes this rank SENT in the last sync()
        self.last_wire_bytes_by_param: Dict[int, int] = {}  # ... per parameter index (large / level tables), "small" = -1
        self.last_list_levels: Dict[int, List[int]] = {}
        # profile=True: device events around sync() and at the first hook-started exchange of every step -> timing()
        self.profile = profile
        self._ev_first_hook = None
        self._ev_steps: List[Tuple[object, object, object]] = []  # (first hook | None, sync entry, sync exit)
        # auto_sync=True: sync() runs by itself at the END of every backward pass (autograd's queue_callback, the mechanism DDP
        # finalises its buckets with) -- for callers that own neither the backward call nor the optimizer step: the reference's
        # trainer goes `grad_scaler.scale(loss).backward()` straight into `grad_scaler.step(optimizer)` (engine/trainer.py:553-
        # 558), and the GradScaler's inf check must already see the REDUCED gradients, or one rank would skip a step the
        # others take.
        self.auto_sync = auto_sync
        self._callback_queued = False
        self.last_sync_bytes = 0
        if auto_sync:
            for p in self.params:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._queue_end_of_backward))
        if overlap:
            for i, p in enumerate(self.params):
                if i in self._levels:  # counts + the previously dense levels start from the hook (_start_level_table)
                    self._hooks.append(p.register_post_accumulate_grad_hook(lambda param, i=i: self._on_level_grad_ready(i)))
                elif self._is_large(p):
                    self._hooks.append(p.register_post_accumulate_grad_hook(lambda param, i=i: self._on_grad_ready(i)))

    def world_size(self) -> int:
        return dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1

    def _queue_end_of_backward(self, _param) -> None:
        """first gradient of a backward pass -> one callback at the end of that pass"""
        if self._callback_queued or self.world_size() == 1:
            return
        self._callback_queued = True
        torch.autograd.Variable._execution_engine.queue_callback(self._end_of_backward)

    def _end_of_backward(self) -> None:
        self._callback_queued = False
        self.last_sync_bytes = self.sync()

    def _is_large(self, t: Tensor) -> bool:
        n = t.numel()
        return n * t.element_size() >= self.large_threshold and t.is_contiguous() and n % max(self.world_size(), 1) == 0

    # ---- which parameters take part -------------------------------------------------------------------------------
    def _local_pattern(self) -> List[bool]:
        return [p.grad is not None for p in self.params]

    def _agree(self, local: List[bool]) -> Tuple[List[bool], List[bool]]:
        """-> (has a gradient on ANY rank, has a gradient on EVERY rank): one tiny MAX all-reduce over (flag, 1 - flag)"""
        dev = self.params[0].device
        m = torch.tensor([int(u) for u in local] + [int(not u) for u in local], device=dev, dtype=torch.int32)
        dist.all_reduce(m, op=dist.ReduceOp.MAX, group=self.group)
        v, n = m.tolist(), len(local)
        return [bool(x) for x in v[:n]], [not bool(x) for x in v[n:]]

    def _used_mask(self) -> List[bool]:
        """Which parameters have a gradient on ANY rank (missing ones are treated as zeros)."""
        local = self._local_pattern()
        if self.usage == "dynamic":
            return self._agree(local)[0]
        if self._agreed is None:
            (self._agreed, self._agreed_all), self._local_at_agreement = self._agree(local), local
        elif local != self._local_at_agreement:
            raise RuntimeError("GradientSynchronizer(usage='static'): the set of parameters with a gradient changed on "
                               "this rank; use usage='dynamic' for models whose used parameters vary (actor grids)")
        return self._agreed

    # ---- the exchange ----------------------------------------------------------------------------------------------
    def _start_large(self, i: int, async_op: bool):
        g = self.params[i].grad
        flat = g.view(-1)
        if self._wire16(g):
            work, recv = scatter_16bit_start(flat, self.world_size(), self.group, async_op, self.wire_dtype)
            if not async_op:
                self._finish_large(flat, scatter_16bit_finish(recv, self.wire_dtype))
                return
            self._inflight[i] = (work, flat, recv, True)
            return
        work, shard = reduce_scatter_flat(flat, self.world_size(), self.group, async_op)
        if not async_op:
            self._finish_large(flat, shard)
            return
        self._inflight[i] = (work, flat, shard, False)

    def _wire16(self, g: Tensor) -> bool:
        return self.wire_dtype is not None and g.dtype == torch.float32

    def _finish_large(self, flat: Tensor, shard: Optional[Tensor], wire16: bool = False) -> None:
        if wire16:  # hook-started 16-bit scatter (tagged in _inflight): `shard` is the receive buffer, the owner's fp32 sum first
            shard = scatter_16bit_finish(shard, self.wire_dtype)
        if shard is None:  # the backend had no reduce-scatter for this device: `flat` already holds the full sum
            if self.average:
                flat.div_(self.world_size())
            return
        if self.average:
            shard.div_(self.world_size())
        all_gather_flat(flat, shard, self.world_size(), self.group)

    # ---- level-sparse exchange ----------------------------------------------------------------------------------------
    @staticmethod
    def _row_mask(g: Tensor, n_levels: int) -> Tensor:
        return (g.view(n_levels, -1, g.shape[-1]) != 0).any(-1)  # [levels, T]

    def _level_counts(self, g: Tensor, n_levels: int):
        """-> (non-zero rows per level, int64 [levels] on g's device; what the compaction needs later).  GPU gradients: two
        launches of csrc/grad_rows.hip (count + per-block prefix); CPU tensors (the gloo tests): torch ops."""
        if g.is_cuda:
            from .. import ops

            return ops.grad_rows_count(g, n_levels)
        mask = self._row_mask(g, n_levels)
        return mask.sum(1).to(torch.int64), mask

    def _start_level_table(self, i: int, early: bool) -> None:
        """Count the rows, start the MAX all-reduce of the counts (tiny, async) and -- ``early``: called from the gradient's
        hook -- the dense reduce-scatter of the levels that went densely in the PREVIOUS step (that set is agreed: it was
        derived from all-reduced counts), so that the bulk of a table's exchange runs under the rest of the backward exactly
        as for a plain large gradient.  Which of the remaining levels go as lists is decided in sync(), from this step's counts."""
        g, L = self.params[i].grad, self._levels[i]
        T, F = g.shape[0] // L, g.shape[1]
        world = self.world_size()
        counts, aux = self._level_counts(g, L)
        cw = dist.all_reduce(counts, op=dist.ReduceOp.MAX, group=self.group, async_op=True)
        runs = []
        prev = self._dense_prev.get(i) if early else None
        if prev and (T * F) % world == 0:
            flat, l = g.view(-1), 0
            while l < L:  # maximal runs of previously dense levels
                if l not in prev:
                    l += 1
                    continue
                e = l
                while e < L and e in prev:
                    e += 1
                run = flat[l * T * F:e * T * F]
                work, shard = reduce_scatter_flat(run, world, self.group, async_op=True)
                runs.append((l, e, work, run, shard))
                l = e
        self._inflight_levels[i] = (counts, cw, aux, runs)

    def _on_level_grad_ready(self, i: int) -> None:
        """post-accumulate-grad hook of a level table (overlap): same conditions as ``_on_grad_ready``"""
        if self._agreed is None or not self._agreed_all[i] or self.world_size() == 1:
            return
        g = self.params[i].grad
        if g.dtype != torch.float32 or not g.is_contiguous():  # (sync() sends such a table like any other gradient)
            return
        if i in self._inflight_levels:
            raise RuntimeError("GradientSynchronizer(overlap=True): a second backward reached level table "
                               f"{i} before sync(); call sync() after every backward (gradient accumulation: overlap=False)")
        if self.profile and self._ev_first_hook is None and self.params[i].is_cuda:
            self._ev_first_hook = torch.cuda.Event(enable_timing=True)
            self._ev_first_hook.record()
        self._start_level_table(i, early=True)

    def _sync_level_tables(self, todo: List[int]) -> None:
        """Hash-table gradients whose coarse levels go as (row, values) lists.  One MAX all-reduce of the per-level row counts
        per table and ONE host read for all of them size the lists; lists are padded to the agreed count with row -1."""
        world = self.world_size()
        for i in todo:  # tables whose hook did not run (first step, overlap off): the same collectives, issued now
            if i not in self._inflight_levels:
                self._start_level_table(i, early=False)
        state = {i: self._inflight_levels.pop(i) for i in todo}
        for i in todo:
            state[i][1].wait()
        counts, off = torch.cat([state[i][0] for i in todo]).tolist(), 0
        for i in todo:
            g, L = self.params[i].grad, self._levels[i]
            T, F = g.shape[0] // L, g.shape[1]
            _, _, aux, runs = state[i]
            cap = counts[off:off + L]
            off += L
            started = {l for (a, e, _, _, _) in runs for l in range(a, e)}
            # a list costs (N-1) * cap * (4 + 4F) bytes per rank, the dense pair 2 (N-1)/N * T * 4F
            pays = [cap[l] * (1 + F) * world < 2 * T * F and (T * F) % world == 0 for l in range(L)]
            lists = [l for l in range(L) if pays[l] and l not in started]
            self.last_list_levels[i] = lists
            # next step's early-started dense set from THIS step's agreed counts alone -- not from what was started: a level
            # that went dense once (a spike step) returns to the list exchange as soon as its row count says so
            self._dense_prev[i] = {l for l in range(L) if not pays[l]}
            for (a, e, work, run, shard) in runs:  # started from the hook: wait for the scatter, finish with the gather
                work.wait()
                self._finish_large(run, shard)
                self.last_wire_bytes += 2 * (world - 1) * run.numel() * 4 // world
            self.overlapped_level_runs_last_step += len(runs)
            flat = g.view(-1)
            l = 0
            while l < L:  # maximal runs of the remaining dense levels: reduce-scatter + all-gather in place
                if l in lists or l in started:
                    l += 1
                    continue
                e = l
                while e < L and e not in lists and e not in started:
                    e += 1
                run = flat[l * T * F:e * T * F]
                if run.numel() % world:
                    dist.all_reduce(run, op=dist.ReduceOp.SUM, group=self.group)
                    if self.average:
                        run.div_(world)
                else:
                    _, shard = reduce_scatter_flat(run, world, self.group)
                    self._finish_large(run, shard)
                self.last_wire_bytes += 2 * (world - 1) * run.numel() * 4 // world
                l = e
            caps = [cap[l] for l in lists]
            total = sum(caps)
            if not lists or total == 0:
                continue
            self.last_wire_bytes += (world - 1) * total * (4 + 4 * F)
            if g.is_cuda:  # csrc/grad_rows.hip: ordered compaction, then every rank's list applied in RANK order
                from .. import ops

                rows, vals = ops.grad_rows_compact(g, L, aux, lists, caps, scale=1.0 / world if self.average else 1.0)
                all_rows, all_vals = rows.new_empty((world * total,)), vals.new_empty((world * total, F))
                all_gather_flat(all_rows, rows, world, self.group)
                all_gather_flat(all_vals.view(-1), vals.view(-1), world, self.group)
                ops.grad_rows_apply(g, L, lists, caps, rows, None, add=False)  # own rows -> 0, then the sums in rank order
                for k in range(world):
                    ops.grad_rows_apply(g, L, lists, caps, all_rows[k * total:(k + 1) * total],
                                        all_vals[k * total:(k + 1) * total], add=True)
                continue
            masks = aux
            rows = torch.full((total,), -1, device=g.device, dtype=torch.int32)
            vals = torch.zeros((total, F), device=g.device, dtype=g.dtype)
            ar = torch.arange(T, device=g.device, dtype=torch.int32)
            o = 0
            for l in lists:  # ordered compaction without a host read: the capacity is known, a dump slot takes the rest
                c = cap[l]
                if c:
                    m = masks[l]
                    slot = torch.where(m, m.cumsum(0) - 1, c)
                    buf = torch.full((c + 1,), -1, device=g.device, dtype=torch.int32)
                    buf.scatter_(0, slot, ar)
                    rows[o:o + c] = buf[:c]
                    o += c
            gl = g.view(L, T, F)
            o = 0
            for l in lists:
                c = cap[l]
                if c:
                    r = rows[o:o + c]
                    vals[o:o + c] = gl[l][r.clamp(min=0).long()] * (r >= 0).unsqueeze(1)
                    o += c
            if self.average:
                vals.mul_(1.0 / world)
            all_rows = [torch.empty_like(rows) for _ in range(world)]
            all_vals = [torch.empty_like(vals) for _ in range(world)]
            dist.all_gather(all_rows, rows, group=self.group)
            dist.all_gather(all_vals, vals, group=self.group)
            o = 0
            for l in lists:
                c = cap[l]
                if not c:
                    continue
                # own rows -> 0 (a padded -1 clears row 0 only if row 0 is not in the list, i.e. is zero already), then
                # every rank's list in RANK order: all replicas form the same sums in the same order
                gl[l].index_fill_(0, rows[o:o + c].clamp(min=0).long(), 0.0)
                for k in range(world):
                    r = all_rows[k][o:o + c]
                    gl[l].index_add_(0, r.clamp(min=0).long(), all_vals[k][o:o + c] * (r >= 0).unsqueeze(1))
                o += c

    def _on_grad_ready(self, i: int) -> None:
        """post-accumulate-grad hook (overlap): autograd is done with this gradient -> its reduce-scatter starts now, on
        the backend's own stream, while the rest of the backward keeps the compute stream busy.  Only after the usage
        set has been agreed (first step runs without overlap) and only for parameters EVERY rank holds a local gradient
        for: a rank without one never fires this hook and would issue the collective later, from sync(), in index order --
        the ranks' collective sequences would differ (a hang, or a silent mix-up of equal-sized tables)."""
        if self._agreed is None or not self._agreed_all[i] or self.world_size() == 1:
            return
        if i in self._inflight:
            raise RuntimeError("GradientSynchronizer(overlap=True): a second backward reached parameter "
                               f"{i} before sync(); call sync() after every backward (gradient accumulation: overlap=False)")
        if self.profile and self._ev_first_hook is None and self.params[i].is_cuda:
            self._ev_first_hook = torch.cuda.Event(enable_timing=True)
            self._ev_first_hook.record()
        self._start_large(i, async_op=True)

    @torch.no_grad()
    def sync(self) -> int:
        """All-reduce every gradient; returns the number of payload bytes exchanged per rank."""
        world = self.world_size()
        if world == 1 or not self.params:
            return 0
        ev_in = None
        if self.profile and self.params[0].is_cuda:
            ev_in = torch.cuda.Event(enable_timing=True)
            ev_in.record()
        used = self._used_mask()
        small, nbytes, level_todo = [], 0, []
        self.last_wire_bytes, self.last_list_levels, self.last_wire_bytes_by_param = 0, {}, {}
        self.overlapped_last_step = len(self._inflight)
        self.overlapped_level_runs_last_step = 0
        for i, (p, u) in enumerate(zip(self.params, used)):
            if not u:
                continue
            if p.grad is None:
                p.grad = torch.zeros_like(p)
            g = p.grad
            nbytes += g.numel() * g.element_size()
            if i in self._levels and g.dtype == torch.float32 and g.is_contiguous():
                level_todo.append(i)
                continue
            if i in self._inflight or self._is_large(g):
                per_elt = (2 + g.element_size()) if self._wire16(g) else 2 * g.element_size()  # scatter leg + gather leg
                self.last_wire_bytes_by_param[i] = (world - 1) * g.numel() * per_elt // world
                self.last_wire_bytes += self.last_wire_bytes_by_param[i]
            if i in self._inflight:  # started from the hook: wait for the scatter, finish with the gather
                work, flat, shard, wire16 = self._inflight.pop(i)
                work.wait()
                self._finish_large(flat, shard, wire16)
            elif self._is_large(g):
                # reduce-scatter + all-gather in place on the gradient's own storage: every link busy, no staging copy
                self._start_large(i, async_op=False)
            else:
                small.append(g)
        assert not self._inflight, "a hooked gradient was not consumed by sync()"
        assert all(i in level_todo for i in self._inflight_levels), "a hooked level table was not consumed by sync()"
        if level_todo:
            before = self.last_wire_bytes
            self._sync_level_tables(level_todo)
            self.last_wire_bytes_by_param[level_todo[0]] = self.last_wire_bytes - before  # (all level tables together)
        if small:
            flat = torch.cat([g.reshape(-1) for g in small])
            self.last_wire_bytes_by_param[-1] = 2 * (world - 1) * flat.numel() * flat.element_size() // world
            self.last_wire_bytes += 2 * (world - 1) * flat.numel() * flat.element_size() // world
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
            if self.average:
                flat.div_(world)
            off = 0
            for g in small:
                g.copy_(flat[off:off + g.numel()].view_as(g))
                off += g.numel()
        if ev_in is not None:
            ev_out = torch.cuda.Event(enable_timing=True)
            ev_out.record()
            self._ev_steps.append((self._ev_first_hook, ev_in, ev_out))
            self._ev_first_hook = None
            if len(self._ev_steps) > 64:
                self._ev_steps = self._ev_steps[-64:]
        return nbytes

    def timing(self, last: int = 0) -> Dict[str, float]:
        """profile=True: means over the recorded steps (the last ``last`` of them; 0 = all), in ms of the compute stream's
        timeline.  exposed = sync() entry -> exit (waiting for the hook-started exchanges + everything sync() itself
        exchanges: what the step pays); overlap_window = first hook-started exchange -> sync() entry (the part of the
        backward the early exchanges could hide under).  Synchronizes the device."""
        steps = self._ev_steps[-last:] if last else self._ev_steps
        if not steps:
            return {}
        torch.cuda.synchronize()
        exposed = [a.elapsed_time(b) for _, a, b in steps]
        window = [h.elapsed_time(a) for h, a, _ in steps if h is not None]
        out = {"exchange_exposed_ms": sum(exposed) / len(exposed), "steps": len(exposed),
               "hook_started_exchanges_per_step": self.overlapped_last_step,
               "hook_started_level_runs_per_step": self.overlapped_level_runs_last_step}
        if window:
            out["exchange_overlap_window_ms"] = sum(window) / len(window)
        return out

    def remove_hooks(self) -> None:
        for h in self._hooks:
            h.remove()
        self._hooks = []
