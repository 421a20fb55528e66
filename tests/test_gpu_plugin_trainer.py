"""``HipTrainer`` (integration/trainer.py) against the reference's ``Trainer.train_iteration`` (engine/trainer.py:535-579).

The method's trainer defers the schedulers' step of iteration i to iteration i + 1's optimizer step so that the iteration holds
no host read.  Here both iterations drive the SAME plugin model (two copies, same parameters, same batches) through the
reference's Optimizers + GradScaler, with non-finite losses forced in some iterations (GradScaler skips those steps, halves the
scale, and the reference does not step its schedulers then): the learning rate every optimizer step sees, the scale, the
schedulers' and optimizers' step counts and the parameters must be the reference loop's; and the deferred iteration must run
under torch's sync-debug mode without a synchronizing call."""
import dataclasses
import os
import sys
import types
from collections import defaultdict
from copy import deepcopy

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ref_import  # noqa: E402
import test_gpu_plugin_train_loop as L  # noqa: E402
import test_gpu_reference_plugin as t  # noqa: E402
from test_gpu_reference_plugin import ref  # noqa: E402,F401

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not ref_import.reference_available(), reason="no reference (oracle/_ref ships with the lease)")]

K = 12
NON_FINITE_AT = (2, 3, 7)


class _PoisonedPipeline(L._Pipeline):
    """the test pipeline with one loss term made infinite in the chosen iterations; the K batches are built up front (their
    host -> device copies would trip the sync-debug mode the deferred iteration runs under)"""

    def prefetch(self, n):
        self._cache = {}
        for step in range(n):
            rb, lab = L._Pipeline._next_train(self, step)
            self._cache[step] = (rb, lab, self.config.ray_patch_size)
        self.datamanager = types.SimpleNamespace(next_train=self._cached_next_train)

    def _cached_next_train(self, step):
        rb, lab, self.config.ray_patch_size = self._cache[step]
        # (the model scales pixel_area in place: every iteration gets its own)
        return dataclasses.replace(rb, pixel_area=rb.pixel_area.clone(), metadata=dict(rb.metadata)), dict(lab)

    def get_train_loss_dict(self, step):
        from neurad_studio_amd.integration.pipeline import ADHipPipeline

        out, loss_dict, metrics = ADHipPipeline.get_train_loss_dict(self, step)
        if step in NON_FINITE_AT:
            loss_dict["rgb_loss"] = loss_dict["rgb_loss"] * float("inf")
        return out, loss_dict, metrics


def _make_loop(cls, method, model, device="cuda:0"):
    """an instance of the trainer class with the attributes Trainer.__init__ / Trainer.setup would give it
    (engine/trainer.py:176-189,264-275) -- no experiment directory, viewer or data"""
    from nerfstudio.engine.optimizers import Optimizers
    from nerfstudio.engine.trainer import Trainer
    from torch.cuda.amp.grad_scaler import GradScaler

    from neurad_studio_amd.optim import TableGradScaler

    loop = object.__new__(cls)
    loop.config = types.SimpleNamespace(log_gradients=False, deferred_scheduler_step=True)
    loop.device, loop.mixed_precision = device, True
    # (engine/trainer.py:189; HipTrainer.__init__ puts its read-only-inf-check scaler there)
    loop.grad_scaler = TableGradScaler(enabled=True) if cls is not Trainer else GradScaler(enabled=True)
    loop.gradient_accumulation_steps = defaultdict(lambda: 1)
    loop.pipeline = _PoisonedPipeline(model, False, "cuda", torch.float32)
    loop.pipeline.prefetch(K)
    groups = {k: v for k, v in model.get_param_groups().items() if len(v)}
    table = deepcopy({k: method.optimizers[k] for k in groups})
    for v in table.values():  # no warm-up, a decay that moves the rate visibly per scheduler step
        v["scheduler"].warmup_steps, v["scheduler"].max_steps = 0, 40
    loop.optimizers = Optimizers(table, groups)
    loop.seen_lr = defaultdict(list)
    for name, opt in loop.optimizers.optimizers.items():
        opt.register_step_pre_hook(lambda o, a, k, name=name: loop.seen_lr[name].append(float(o.param_groups[0]["lr"])))
    return loop


def test_deferred_scheduler_step_is_the_reference_iteration(ref):
    from nerfstudio.engine.trainer import Trainer

    from neurad_studio_amd.integration.trainer import HipTrainer, HipTrainerConfig

    methods = L._methods()
    method = methods["neurad-hip"]
    assert isinstance(method, HipTrainerConfig) and method._target is HipTrainer  # what `ns-train neurad-hip` instantiates
    a, _ = t._build_pair(ref, False)
    twins = {"reference": a, "reference again": t._build_pair(ref, False)[0], "deferred": t._build_pair(ref, False)[0]}
    init = {n: p.detach().float().clone() for n, p in a.named_parameters()}
    loops = {}
    for who, m in twins.items():
        m.load_state_dict(a.state_dict())
        t._deterministic(m, True)
        cls = HipTrainer if who == "deferred" else Trainer
        loops[who] = (_make_loop(cls, method, m), cls.train_iteration)
    sync_free = 0
    for step in range(K):
        for who, (loop, iterate) in loops.items():
            strict = who == "deferred" and step >= 2  # (the first iterations allocate: pinned flags, optimizer state)
            if strict:
                torch.cuda.set_sync_debug_mode("error")
            try:
                iterate(loop, step)
                sync_free += strict
            finally:
                torch.cuda.set_sync_debug_mode("default")
            loop.pipeline.model.sampler.step_cb(step)
    ref_loop, hip_loop = loops["reference"][0], loops["deferred"][0]
    hip_loop._settle_schedulers()
    assert sync_free == K - 2
    skipped = len(NON_FINITE_AT)
    assert ref_loop.grad_scaler.get_scale() == hip_loop.grad_scaler.get_scale() == 65536.0 * 0.5 ** skipped
    for name, seen in ref_loop.seen_lr.items():
        assert len(seen) == K and seen == hip_loop.seen_lr[name], (name, seen, hip_loop.seen_lr[name])
        assert len(set(seen)) == K - skipped, (name, seen)  # the rate stood still across the skipped iterations
        sa, sb = ref_loop.optimizers.schedulers[name], hip_loop.optimizers.schedulers[name]
        assert sa.last_epoch == sb.last_epoch == K - skipped and sa.get_last_lr() == sb.get_last_lr()
        oa, ob = ref_loop.optimizers.optimizers[name], hip_loop.optimizers.optimizers[name]
        for pa_, pb_ in zip(oa.param_groups[0]["params"], ob.param_groups[0]["params"]):
            if pa_ in oa.state:
                assert float(oa.state[pa_]["step"]) == float(ob.state[pb_]["step"]) == K - skipped, name

    # The parameters: the weight gradients are sums of float atomics, and Adam at eps = 1e-15 turns their last-bit differences
    # into +- lr steps wherever a gradient is ~ 0 -- two runs of the REFERENCE's iteration differ by that much; the deferred
    # iteration must not differ from the reference's by more.  (distance / the update the reference run made)
    def distance(x, y):
        px, py = dict(x.named_parameters()), dict(y.named_parameters())
        out = {}
        for n, p in px.items():
            moved = float((p.detach().float() - init[n]).norm())
            if moved > 0:
                out[n] = float((p.detach().float() - py[n].detach().float()).norm()) / moved
        return out

    rerun, deferred = distance(a, twins["reference again"]), distance(a, twins["deferred"])
    worst_rerun, worst = max(rerun.values()), max(deferred.values())
    mean_rerun, mean = sum(rerun.values()) / len(rerun), sum(deferred.values()) / len(deferred)
    print(f"parameters after {K} iterations ({skipped} skipped), distance / update over {len(rerun)} tensors (worst, mean): "
          f"the reference's iteration run twice {worst_rerun:.3e}, {mean_rerun:.3e}; deferred vs reference {worst:.3e}, {mean:.3e}")
    assert worst <= 3.0 * worst_rerun + 1e-6, (worst, worst_rerun)
    assert mean <= 2.0 * mean_rerun + 1e-6, (mean, mean_rerun)
