"""Optimizer step of the hash tables on the HIP kernel.

``HashGridAdam`` is torch.optim.Adam / AdamW for large fp32 tables: same hyper-parameters, same arithmetic, same
``state_dict`` layout (``step``, ``exp_avg``, ``exp_avg_sq`` per parameter -- checkpoints written by the reference's
``Optimizers`` wrapper, engine/optimizers.py:168-181, load into it and vice versa), one streaming kernel per large table and
one launch per 24 small ones (the per-actor grids) that skips the rows whose update is provably a no-op (csrc/adam.hip).
fp16-storage tables: the kernel reads the fp16 gradient, updates the fp32 master copy in the optimizer state and writes the
rounded fp16 table in the same pass.  The reference's settings for its ``hashgrids`` group
are ``AdamOptimizerConfig(lr=1e-2, eps=1e-15)`` (configs/method_configs.py:423-426)."""
from __future__ import annotations

from typing import Iterable, Optional, Tuple

import torch

from . import ops


class HashGridAdam(torch.optim.Optimizer):
    def __init__(self, params: Iterable, lr: float = 1e-2, betas: Tuple[float, float] = (0.9, 0.999), eps: float = 1e-15,
                 weight_decay: float = 0.0, decoupled_weight_decay: bool = True) -> None:
        if weight_decay and not decoupled_weight_decay:
            raise NotImplementedError("L2-in-gradient weight decay; use decoupled (AdamW) decay or 0")
        # the remaining keys are torch.optim.Adam's own group entries, carried so that a state_dict saved here loads into
        # torch.optim.Adam / AdamW with the same meaning (and the other way round)
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False, maximize=False,
                                      foreach=None, capturable=False, differentiable=False, fused=None,
                                      decoupled_weight_decay=bool(weight_decay)))

    @torch.no_grad()
    def step(self, closure=None, grad_scale: Optional[float] = None):
        """grad_scale: multiply every gradient by it inside the kernel (1 / GradScaler scale when the caller does not
        want a separate unscale pass over 600 MB of gradients)."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            if group.get("amsgrad") or group.get("maximize") or (group["weight_decay"] and not group.get("decoupled_weight_decay", True)):
                raise NotImplementedError("HashGridAdam: amsgrad / maximize / L2-in-gradient weight decay")
            b1, b2 = group["betas"]
            items = []
            for p in group["params"]:
                if p.grad is None:
                    continue  # untouched tables (an actor no ray hit): state must not decay, as in torch
                st = self.state[p]
                if not st:
                    # the step count is kept as a host int beside torch's 0-d tensor: no tensor arithmetic and no .item() per
                    # table and step (66 tables on a 32-actor scene)
                    st["step"] = torch.tensor(0.0)  # host scalar, like torch.optim.Adam(capturable=False)
                    st["exp_avg"] = torch.zeros_like(p, dtype=torch.float32, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, dtype=torch.float32, memory_format=torch.preserve_format)
                step = int(st["step"]) + 1
                st["step"] = torch.tensor(float(step))  # (a new tensor: a state_dict loaded from a live optimizer shares this scalar)
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
                items.append((target, grad, st["exp_avg"], st["exp_avg_sq"], step, image))
            # one launch per 24 tensors (csrc/adam.hip): the large tables get the machine to themselves in their own launch
            big = [it for it in items if it[0].numel() >= 1 << 24]
            small = [it for it in items if it[0].numel() < 1 << 24]
            for it in big:
                ops.adam_step_many([it], group["lr"], b1, b2, group["eps"], group["weight_decay"],
                                   1.0 if grad_scale is None else grad_scale)
            ops.adam_step_many(small, group["lr"], b1, b2, group["eps"], group["weight_decay"],
                               1.0 if grad_scale is None else grad_scale)
            # the kernel wrote the tables through raw pointers: the parameters' version counters did not move.  Caches keyed
            # on a table's version (ops.eval_table under NRHIP_EVAL_RELAYOUT=1: the re-laid-out coarse levels) would serve
            # the old values to an eval that stays in train mode; dropping them here costs nothing when none exist.
            if items:
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

