"""Renderers with the reference's signatures (model_components/renderers.py:59-90,322-350,353-418) and
render_depth_simple (models/neurad.py:727-734), dense mode, on the HIP compositing kernels."""
from __future__ import annotations

from typing import Optional

import torch
from torch import Tensor, nn

from ..cameras.rays import RaySamples
from ..shims import nerfacc


class FeatureRenderer(nn.Module):
    def forward(self, features: Tensor, weights: Tensor, ray_indices=None, num_rays=None) -> Tensor:
        return nerfacc.accumulate_along_rays(weights[..., 0], values=features, ray_indices=ray_indices, n_rays=num_rays)


class AccumulationRenderer(nn.Module):
    @classmethod
    def forward(cls, weights: Tensor, ray_indices=None, num_rays=None) -> Tensor:
        return nerfacc.accumulate_along_rays(weights[..., 0], values=None, ray_indices=ray_indices, n_rays=num_rays)


def render_depth_simple(weights: Tensor, ray_samples: RaySamples, ray_indices=None, num_rays=None) -> Tensor:
    steps = (ray_samples.frustums.starts + ray_samples.frustums.ends) / 2
    return nerfacc.accumulate_along_rays(weights[..., 0], values=steps, ray_indices=ray_indices, n_rays=num_rays)


class DepthRenderer(nn.Module):
    def __init__(self, method: str = "expected") -> None:
        super().__init__()
        if method != "expected":
            raise NotImplementedError("NeuRAD uses DepthRenderer('expected') only (models/neurad.py:252)")
        self.method = method

    def forward(self, weights: Tensor, ray_samples: RaySamples, ray_indices=None, num_rays=None) -> Tensor:
        eps = 1e-10
        steps = (ray_samples.frustums.starts + ray_samples.frustums.ends) / 2
        depth = nerfacc.accumulate_along_rays(weights[..., 0], values=steps)
        acc = nerfacc.accumulate_along_rays(weights[..., 0], values=None)
        depth = depth / (acc + eps)
        return torch.clip(depth, steps.min(), steps.max())


class NormalsRenderer(nn.Module):
    """Weighted sum of the per-sample normals, optionally renormalised (model_components/renderers.py:462-489; the
    normalisation is safe_normalize, utils/math.py:455-468: v / (|v| + 1e-10))."""

    @classmethod
    def forward(cls, normals: Tensor, weights: Tensor, normalize: bool = True, ray_indices=None, num_rays=None) -> Tensor:
        n = nerfacc.accumulate_along_rays(weights[..., 0], values=normals, ray_indices=ray_indices, n_rays=num_rays)
        if normalize:
            n = n / (torch.linalg.norm(n, dim=-1, keepdim=True) + 1e-10)
        return n
