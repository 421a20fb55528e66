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


class OccGridEstimator:
    """nerfacc.OccGridEstimator-shaped holder of a single-level binary occupancy grid with ``sampling`` as
    VolumetricSampler calls it (model_components/ray_samplers.py:527-540).  Marching rule: csrc/occgrid.hip."""

    def __init__(self, roi_aabb, resolution: int = 128, levels: int = 1, device="cuda"):
        import torch

        if levels != 1:
            raise NotImplementedError("multi-level occupancy grids")
        self.aabbs = torch.as_tensor(roi_aabb, dtype=torch.float32).reshape(1, 6)
        self.binaries = torch.ones((1, resolution, resolution, resolution), dtype=torch.bool, device=device)
        self.occs = self.binaries.float().reshape(-1)

    def sampling(self, rays_o, rays_d, sigma_fn=None, alpha_fn=None, near_plane: float = 0.0, far_plane: float = 1e10,
                 t_min=None, t_max=None, render_step_size: float = 1e-3, early_stop_eps: float = 1e-4,
                 alpha_thre: float = 0.0, stratified: bool = False, cone_angle: float = 0.0):
        import torch

        from .. import ops

        t_rand = torch.rand((rays_o.shape[0],), device=rays_o.device) if stratified else None
        grid = ops.OccGridSpec(self.aabbs[0], self.binaries[0])
        ri, ts, te, seg = ops.occgrid_march(grid, rays_o, rays_d, render_step_size, near_plane, far_plane, t_min, t_max,
                                            cone_angle, t_rand)
        if (alpha_thre > 0.0 or early_stop_eps > 0.0) and (sigma_fn is not None or alpha_fn is not None) and ri.numel():
            alpha_thre = min(alpha_thre, float(self.occs.mean().item()))
            if sigma_fn is not None:
                alphas = 1.0 - torch.exp(-sigma_fn(ts, te, ri) * (te - ts))
            else:
                alphas = alpha_fn(ts, te, ri)
            keep = ops.packed_visibility_from_alpha(alphas.float().contiguous(), seg, early_stop_eps, alpha_thre)
            ri, ts, te = ri[keep], ts[keep], te[keep]
        return ri, ts, te
