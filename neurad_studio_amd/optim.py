"""Optimizer step of the hash tables on the HIP kernel.

``HashGridAdam`` is torch.optim.Adam / AdamW for large fp32 tables: same hyper-parameters, same arithmetic, same
``state_dict`` layout (``step``, ``exp_avg``, ``exp_avg_sq`` per parameter -- checkpoints written by the reference's
``Optimizers`` wrapper, engine/optimizers.py:168-181, load into it and vice versa), one streaming kernel per large table and
one launch per 24 small ones (the per-actor grids) that skips the rows whose update is provably a no-op (csrc/adam.hip).
fp16-storage tables: the kernel reads the fp16 gradient, updates the fp32 master copy in the optimizer state and writes the
rounded fp16 table in the same pass.  The reference's settings for its ``hashgrids`` group
are ``AdamOptimizerConfig(lr=1e-2, eps=1e-15)`` (configs/method_configs.py:423-426).

Under the reference's trainer (engine/trainer.py:550-576: ``torch.autocast`` + ``grad_scaler.step(optimizer)`` for every
group, ``mixed_precision=True`` is the ``neurad`` default) the optimizer speaks torch.amp.GradScaler's protocol for
optimizers that handle the scale themselves (``_step_supports_amp_scaling``, what torch's fused Adam does): GradScaler hands
``self.grad_scale`` / ``self.found_inf`` as device scalars, the kernel divides the gradients by the scale inside the update
and leaves everything untouched -- step counts included -- when an inf was found; no host read, no unscale pass over the
table gradients, and fp16 gradients (fp16-storage tables) are accepted, which ``GradScaler.unscale_`` refuses.
``capturable=True`` keeps the step counts on the device from the first step (torch's capturable layout) so that a whole
training step can be captured in a HIP graph; ``lr`` may then be an fp32 device scalar that a scheduler fills."""
from __future__ import annotations

from typing import Iterable, Optional, Tuple, Union

import torch

from . import ops


class HashGridAdam(torch.optim.Optimizer):
    _step_supports_amp_scaling = True  # torch/amp/grad_scaler.py: GradScaler.step registers grad_scale / found_inf on us

    def __init__(self, params: Iterable, lr: Union[float, torch.Tensor] = 1e-2, betas: Tuple[float, float] = (0.9, 0.999),
                 eps: float = 1e-15, weight_decay: float = 0.0, decoupled_weight_decay: bool = True,
                 capturable: bool = False) -> None:
        if weight_decay and not decoupled_weight_decay:
            raise NotImplementedError("L2-in-gradient weight decay; use decoupled (AdamW) decay or 0")
        # the remaining keys are torch.optim.Adam's own group entries, carried so that a state_dict saved here loads into
        # torch.optim.Adam / AdamW with the same meaning (and the other way round)
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False, maximize=False,
                                      foreach=None, capturable=bool(capturable), differentiable=False, fused=None,
                                      decoupled_weight_decay=bool(weight_decay)))
        self._workspace = {}

    def _ws(self, n: int, device) -> torch.Tensor:
        ws = self._workspace.get(device)
        if ws is None or ws.numel() < ops.adam_workspace_floats(n):
            ws = self._workspace[device] = ops.adam_workspace(max(n, 32), device)
        return ws

    @torch.no_grad()
    def step(self, closure=None, grad_scale: Optional[float] = None):
        """grad_scale (host float): multiply every gradient by it inside the kernel.  GradScaler's device-side scale arrives
        as the attribute ``self.grad_scale`` instead (see the module docstring) and divides."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        amp_scale = getattr(self, "grad_scale", None)  # set and removed by GradScaler.step around this call
        found_inf = getattr(self, "found_inf", None)
        amp = isinstance(amp_scale, torch.Tensor) or isinstance(found_inf, torch.Tensor)
        for group in self.param_groups:
            if group.get("amsgrad") or group.get("maximize") or (group["weight_decay"] and not group.get("decoupled_weight_decay", True)):
                raise NotImplementedError("HashGridAdam: amsgrad / maximize / L2-in-gradient weight decay")
            b1, b2 = group["betas"]
            lr = group["lr"]
            items, on_device = [], amp or bool(group.get("capturable")) or isinstance(lr, torch.Tensor)
            for p in group["params"]:
                if p.grad is None:
                    continue  # untouched tables (an actor no ray hit): state must not decay, as in torch
                st = self.state[p]
                if not st:
                    # capturable=False: a host scalar like torch.optim.Adam's -- no tensor arithmetic and no .item() per table
                    # and step (66 tables on a 32-actor scene); device form: torch's capturable layout, an fp32 device scalar
                    st["step"] = torch.zeros((), dtype=torch.float32, device=p.device) if on_device else torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p, dtype=torch.float32, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, dtype=torch.float32, memory_format=torch.preserve_format)
                on_device = on_device or st["step"].is_cuda  # a count that lives on the device stays there (no host read)
                image = None
                target = p
                if p.dtype != torch.float32:
                    # fp16-storage table (BASELINE config 5): the update runs on an fp32 master copy kept in the
                    # optimizer state (what tiny-cuda-nn does for its fp16 parameters); the table is its rounded image,
                    # written by the same kernel, which also reads the fp16 gradient as it is (no .float() / copy_ passes).
                    # Checked by DTYPE, not by key presence: a state that went through a generic load_state_dict may
                    # hold these tensors cast to the parameter's dtype (torch does that to floating-point state).
                    if "master" not in st or st["master"].dtype != torch.float32:
                        st["master"] = p.detach().float()
                    for k in ("exp_avg", "exp_avg_sq"):
                        if st[k].dtype != torch.float32:
                            st[k] = st[k].float()
                    target, image = st["master"], p.detach()
                grad = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                if grad.dtype not in (torch.float32, torch.float16):
                    grad = grad.float()
                items.append((target, grad, st["exp_avg"], st["exp_avg_sq"], st, image))
            if not items:
                continue
            host_scale = 1.0 if grad_scale is None else grad_scale
            if on_device:
                dev_items = []
                for (target, grad, m, v, st, image) in items:
                    if not st["step"].is_cuda or st["step"].dtype != torch.float32:  # (a host count from an earlier mode / a checkpoint)
                        st["step"] = st["step"].detach().to(device=target.device, dtype=torch.float32).reshape(())
                    dev_items.append((target, grad, m, v, st["step"], image))
                dev = dev_items[0][0].device
                if isinstance(lr, torch.Tensor) and (not lr.is_cuda or lr.dtype != torch.float32):
                    lr = float(lr)
                scale_t = amp_scale if isinstance(amp_scale, torch.Tensor) else None
                inf_t = found_inf if isinstance(found_inf, torch.Tensor) else None
                if scale_t is not None and (scale_t.dtype != torch.float32 or scale_t.device != dev):
                    scale_t = scale_t.to(device=dev, dtype=torch.float32)
                if inf_t is not None and (inf_t.dtype != torch.float32 or inf_t.device != dev):
                    inf_t = inf_t.to(device=dev, dtype=torch.float32)
                ws = self._ws(len(dev_items), dev)
                # the large tables get the machine to themselves in their own launch (csrc/adam.hip); each launch takes its own
                # slice of the workspace
                big = [it for it in dev_items if it[0].numel() >= 1 << 24]
                small = [it for it in dev_items if it[0].numel() < 1 << 24]
                off = 0
                for batch in [[it] for it in big] + ([small] if small else []):
                    n_floats = ops.adam_workspace_floats(len(batch))
                    ops.adam_step_many_dev(batch, lr, b1, b2, group["eps"], group["weight_decay"], scale_t, inf_t,
                                           ws[off:off + n_floats], host_scale)
                    off += n_floats
            else:
                host_items = []
                for (target, grad, m, v, st, image) in items:
                    step = int(st["step"]) + 1
                    st["step"] = torch.tensor(float(step))  # (a new tensor: a state_dict loaded from a live optimizer shares this scalar)
                    host_items.append((target, grad, m, v, step, image))
                # one launch per 24 tensors (csrc/adam.hip): the large tables get the machine to themselves in their own launch
                big = [it for it in host_items if it[0].numel() >= 1 << 24]
                small = [it for it in host_items if it[0].numel() < 1 << 24]
                for it in big:
                    ops.adam_step_many([it], float(lr), b1, b2, group["eps"], group["weight_decay"], host_scale)
                ops.adam_step_many(small, float(lr), b1, b2, group["eps"], group["weight_decay"], host_scale)
            # the kernel wrote the tables through raw pointers: the parameters' version counters did not move.  Caches keyed
            # on a table's version (ops.eval_table under NRHIP_EVAL_RELAYOUT=1: the re-laid-out coarse levels) would serve
            # the old values to an eval that stays in train mode; dropping them here costs nothing when none exist.
            ops.clear_eval_tables()
        return loss

    def load_state_dict(self, state_dict) -> None:
        """torch.optim.Optimizer.load_state_dict casts every floating-point state tensor to its parameter's dtype; for an
        fp16-storage table that would round the fp32 master copy and both moments to fp16.  They are restored from the
        tensors of ``state_dict`` itself, in fp32."""
        super().load_state_dict(state_dict)
        saved = state_dict["state"]
        ids = [i for g in state_dict["param_groups"] for i in g["params"]]
        params = [p for g in self.param_groups for p in g["params"]]
        for i, p in zip(ids, params):
            if p.dtype == torch.float32 or i not in saved:
                continue
            for k in ("master", "exp_avg", "exp_avg_sq"):
                if k in saved[i]:
                    self.state[p][k] = saved[i][k].detach().to(device=p.device, dtype=torch.float32).clone()


class TableGradScaler(torch.amp.GradScaler):
    """torch.amp.GradScaler whose inf check over a ``HashGridAdam``'s gradients only READS them.

    ``GradScaler.step`` runs ``_check_inf_per_device`` on every optimizer that consumes the scale itself
    (torch/amp/grad_scaler.py; the reference steps each group through it, engine/optimizers.py:168-181): torch's
    ``_amp_foreach_non_finite_check_and_unscale_`` at a scale of 1 reads AND re-writes every gradient element -- 1.2 GB per
    step for NeuRAD's 600 MB of table gradients.  For a HashGridAdam the same flag comes from ``ops.nonfinite_check`` (one
    read); every other optimizer, and any gradient that kernel does not take (sparse, other devices, unaligned views), goes
    through torch's own pass.  Scale growth / backoff, ``unscale_``, ``state_dict``: inherited unchanged.  (The override point
    is the one torch's own ShardedGradScaler uses.)"""

    def __init__(self, device: str = "cuda", **kwargs) -> None:
        super().__init__(device, **kwargs)

    def _check_inf_per_device(self, optimizer):
        if isinstance(optimizer, HashGridAdam):
            grads = [p.grad for g in optimizer.param_groups for p in g["params"] if p.grad is not None]
            devices = {g.device for g in grads}
            if grads and len(devices) == 1 and all(
                    g.is_cuda and not g.is_sparse and g.dtype in (torch.float32, torch.float16) and g.is_contiguous()
                    and g.data_ptr() % 16 == 0 for g in grads):
                self._check_scale_growth_tracker("_check_inf_per_device")
                device = grads[0].device
                found_inf = torch.zeros((), dtype=torch.float32, device=device)
                ops.nonfinite_check(grads, found_inf)
                state = self._per_optimizer_states[id(optimizer)]
                state["found_inf_per_device"] = {device: found_inf}
                return state["found_inf_per_device"]
        return super()._check_inf_per_device(optimizer)
