"""Trainer of the ``neurad-hip`` method: the reference's iteration without its two host reads.

``Trainer.train_iteration`` (engine/trainer.py:535-579) ends with

    scale = grad_scaler.get_scale(); grad_scaler.update(); if scale <= grad_scaler.get_scale(): schedulers step

-- two ``.item()`` reads per iteration, each of which drains the device queue: the host cannot enqueue iteration i + 1 while
iteration i runs.  With the reference's torch model that hides behind its ~100 ms step; behind the HIP step (8.9 ms of device
time, 7.3 ms of host enqueue) it is ~0.8 ms of idle device per iteration (profiles/r06_via_plugin_c3_gpu_gaps.txt).

``HipTrainer.train_iteration`` runs the same calls in the same order with ONE change of schedule: whether the scale went down
in iteration i is computed on the device (``grad_scaler.scale(1)`` before and after ``update()``: public API, no read), copied
to pinned memory, and the schedulers' step of iteration i is taken when iteration i + 1 reaches its optimizer step -- the first
place a learning rate is read -- by which time the copy finished long ago.  Every optimizer step therefore sees the learning
rate the reference's loop would give it (tests/test_gpu_plugin_trainer.py forces skipped iterations and compares both loops);
``save_checkpoint`` settles the pending step first, so a checkpoint holds the scheduler state the reference's would."""
from __future__ import annotations

import functools
from dataclasses import dataclass, field
from typing import Type

import torch

from nerfstudio.engine.trainer import Trainer, TrainerConfig

from ..optim import TableGradScaler


@dataclass
class HipTrainerConfig(TrainerConfig):
    _target: Type = field(default_factory=lambda: HipTrainer)
    deferred_scheduler_step: bool = True
    """the scale-went-down test of iteration i is read at iteration i + 1's optimizer step (False: the reference's two
    ``get_scale()`` reads per iteration)"""
    read_only_inf_check: bool = True
    """``optim.TableGradScaler`` in place of torch's GradScaler: the inf check over the table gradients reads them once
    instead of reading and re-writing them"""


class HipTrainer(Trainer):
    config: HipTrainerConfig

    def __init__(self, config: HipTrainerConfig, local_rank: int = 0, world_size: int = 1) -> None:
        super().__init__(config, local_rank, world_size)
        if config.read_only_inf_check and str(self.device).startswith("cuda"):
            # (engine/trainer.py:188 builds torch's scaler; checkpoints load into whichever is here, engine/trainer.py:455)
            self.grad_scaler = TableGradScaler(enabled=self.use_grad_scaler)

    def _settle_schedulers(self) -> None:
        """take the schedulers' step a finished iteration left pending (a no-op when none is)"""
        pending = getattr(self, "_pending_scheduler_step", None)
        if pending is None:
            return
        self._pending_scheduler_step = None
        step, went_down, copied = pending
        copied.synchronize()
        if not bool(went_down.item()):  # (pinned host memory: a plain read)
            self.optimizers.scheduler_step_all(step)

    def train_iteration(self, step: int):
        deferred = getattr(self.config, "deferred_scheduler_step", True) and self.grad_scaler.is_enabled() \
            and str(self.device).startswith("cuda")
        if not deferred:
            self._settle_schedulers()
            return super().train_iteration(step)

        groups = list(self.optimizers.parameters.keys())
        accumulate = self.gradient_accumulation_steps
        self.optimizers.zero_grad_some([g for g in groups if step % accumulate[g] == 0])
        with torch.autocast(device_type="cuda", enabled=self.mixed_precision):
            _, loss_dict, metrics_dict = self.pipeline.get_train_loss_dict(step=step)
            loss = functools.reduce(torch.add, loss_dict.values())
        self.grad_scaler.scale(loss).backward()
        self._settle_schedulers()  # iteration step - 1's learning-rate step, before the rates are read
        self.optimizers.optimizer_scaler_step_some(self.grad_scaler, [g for g in groups if step % accumulate[g] == accumulate[g] - 1])

        if self.config.log_gradients:
            total = 0
            for tag, value in self.pipeline.model.named_parameters():
                assert tag != "Total"
                if value.grad is not None:
                    norm = value.grad.norm()
                    metrics_dict[f"Gradients/{tag}"] = norm
                    total = total + norm
            metrics_dict["Gradients/Total"] = total

        if getattr(self, "_scale_probe", None) is None:
            self._scale_probe = torch.ones((), device=self.device)
            self._went_down_host = [torch.zeros((), dtype=torch.bool).pin_memory() for _ in range(2)]
        before = self.grad_scaler.scale(self._scale_probe)  # the scale as a device scalar
        self.grad_scaler.update()
        went_down = self._went_down_host[step & 1]
        went_down.copy_(self.grad_scaler.scale(self._scale_probe) < before, non_blocking=True)
        copied = torch.cuda.Event()
        copied.record()
        self._pending_scheduler_step = (step, went_down, copied)
        return loss, loss_dict, metrics_dict

    def save_checkpoint(self, step: int) -> None:
        self._settle_schedulers()
        super().save_checkpoint(step)
