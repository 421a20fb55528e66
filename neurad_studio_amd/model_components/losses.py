"""Losses that consume the sampler outputs every training step (mirror of the two functions NeuRADModel calls,
nerfstudio/model_components/losses.py:107-112,137-156,645-705; call sites models/neurad.py:262,524,541-545).

Same signatures: ``(weights_list, ray_samples_list) -> scalar``.  One wavefront per ray and per proposal level instead
of ~40 torch kernels; gradients reach what the reference's autograd reaches (the proposal weights for the interlevel
loss -- the fine histogram is detached there --, the final weights for the distortion loss)."""
from __future__ import annotations

from typing import List, Sequence

import torch
from torch import Tensor

from .. import autograd as ag
from ..cameras.rays import RaySamples

PULSE_WIDTHS = (0.03, 0.003)  # losses.py:677


def ray_samples_to_sdist(ray_samples: RaySamples) -> Tensor:
    """spacing-space bin edges [R, S+1] (losses.py:107-112)"""
    sdist = getattr(ray_samples, "sdist", None)
    if sdist is not None:  # the fused training path carries the edges as one tensor
        return sdist
    return torch.cat([ray_samples.spacing_starts[..., 0], ray_samples.spacing_ends[..., -1:, 0]], dim=-1)


def zipnerf_interlevel_loss(weights_list: Sequence[Tensor], ray_samples_list: List[RaySamples]) -> Tensor:
    c = ray_samples_to_sdist(ray_samples_list[-1]).detach()
    w = weights_list[-1].squeeze(-1).detach()
    loss = None
    for i, (ray_samples, weights) in enumerate(zip(ray_samples_list[:-1], weights_list[:-1])):
        term = ag.InterlevelLossFn.apply(c, w, ray_samples_to_sdist(ray_samples).detach(), weights.squeeze(-1), PULSE_WIDTHS[i])
        loss = term if loss is None else loss + term
    return c.new_zeros(()) if loss is None else loss


def distortion_loss(weights_list: Sequence[Tensor], ray_samples_list: List[RaySamples]) -> Tensor:
    return ag.DistortionLossFn.apply(ray_samples_to_sdist(ray_samples_list[-1]).detach(), weights_list[-1].squeeze(-1))
