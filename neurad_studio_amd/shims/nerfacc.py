"""nerfacc-shaped module (dense mode) on HIP kernels: the three functions neurad-studio calls
(models/neurad.py:716-723,734; model_components/renderers.py:88,130,133,345,404,407,455,486).
Put this directory on sys.path as ``nerfacc`` or import it directly."""
from __future__ import annotations

from typing import Optional, Tuple

from torch import Tensor

from .. import autograd as ag


def render_weight_from_alpha(alphas: Tensor, packed_info=None, ray_indices=None, n_rays=None,
                             prefix_trans=None) -> Tuple[Tensor, Tensor]:
    if packed_info is not None or ray_indices is not None:
        raise NotImplementedError("packed mode has no caller in neurad-studio (SURVEY §2.1)")
    return ag.WeightFromAlphaFn.apply(alphas.contiguous())


def render_weight_from_density(t_starts: Tensor, t_ends: Tensor, sigmas: Tensor, packed_info=None, ray_indices=None,
                               n_rays=None, prefix_trans=None) -> Tuple[Tensor, Tensor, Tensor]:
    if packed_info is not None or ray_indices is not None:
        raise NotImplementedError("packed mode has no caller in neurad-studio (SURVEY §2.1)")
    return ag.WeightFromDensityFn.apply(t_starts.contiguous(), t_ends.contiguous(), sigmas.contiguous())


def accumulate_along_rays(weights: Tensor, values: Optional[Tensor] = None, ray_indices: Optional[Tensor] = None,
                          n_rays: Optional[int] = None) -> Tensor:
    if ray_indices is not None:
        raise NotImplementedError("packed mode has no caller in neurad-studio (SURVEY §2.1)")
    if values is None:
        return weights.sum(-1, keepdim=True) if weights.requires_grad else ag.ops.accumulate_along_rays(weights.contiguous())
    return ag.AccumulateFn.apply(weights.contiguous(), values.contiguous())
