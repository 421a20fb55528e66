"""Data contract of the hot path: RayBundle / RaySamples / Frustums.

Mirrors nerfstudio/cameras/rays.py:33-357 (field names, shapes, ``get_ray_samples`` making the per-ray fields
stride-0 broadcast views).  Only what the hot path touches is reproduced; the HIP kernels never consume the
materialised [R,S,3] views -- the fields read per-ray data through ``per_ray()`` and the [R,S] starts/ends.
"""
from __future__ import annotations

from dataclasses import dataclass, field, replace
from typing import Callable, Dict, Optional

import torch
from torch import Tensor

from .. import autograd as ag


def per_ray(frustums):
    """(origins [R,3], directions [R,3], pixel_area [R]) of a [R,S] frustum batch -- of THIS package's Frustums or of
    the reference's (nerfstudio/cameras/rays.py:33-59): both hold the per-ray fields as stride-0 broadcast views
    [R,1,C] -> [R,S,C] (rays.py:336-355), so column 0 is the ray's value and no copy is made for the common case."""
    o, d, a = frustums.origins, frustums.directions, frustums.pixel_area
    if o.dim() == 3:
        o, d, a = o[:, 0], d[:, 0], a[:, 0]
    elif o.dim() != 2:
        raise ValueError(f"frustums must be [R,S,*] (or per-ray [R,*]); got origins of shape {tuple(o.shape)}")
    return o.contiguous(), d.contiguous(), a.reshape(-1).contiguous()


def sample_times(ray_samples):
    """per-ray times [R] of a [R,S] RaySamples (times are broadcast over the samples like the other ray fields)"""
    t = ray_samples.times
    if t is None:
        return None
    return (t[:, 0] if t.dim() == 3 else t).reshape(-1)


def get_weights(ray_samples, densities: Tensor) -> Tensor:
    """RaySamples.get_weights (rays.py:188-210) for any RaySamples-shaped object, on the GPU: wave-per-ray
    exclusive-sum scan (nrhip_weights_from_density)."""
    w = ag.WeightsFromDensityFn.apply(ray_samples.deltas[..., 0].contiguous(), densities[..., 0].contiguous())
    return w[..., None]


@dataclass
class Frustums:
    origins: Tensor      # [*bs, 3]
    directions: Tensor   # [*bs, 3]
    starts: Tensor       # [*bs, 1]
    ends: Tensor         # [*bs, 1]
    pixel_area: Tensor   # [*bs, 1]
    offsets: Optional[Tensor] = None

    def get_positions(self) -> Tensor:
        """rays.py:61-72"""
        pos = self.origins + self.directions * (self.starts + self.ends) / 2
        return pos if self.offsets is None else pos + self.offsets

    @property
    def shape(self):
        return self.starts.shape[:-1]

    def per_ray(self):
        return per_ray(self)


@dataclass
class RaySamples:
    frustums: Frustums
    camera_indices: Optional[Tensor] = None
    deltas: Optional[Tensor] = None            # [*bs, 1]
    spacing_starts: Optional[Tensor] = None    # [*bs, 1]
    spacing_ends: Optional[Tensor] = None
    spacing_to_euclidean_fn: Optional[Callable] = None
    metadata: Optional[Dict[str, Tensor]] = None
    times: Optional[Tensor] = None
    sdist: Optional[Tensor] = None
    """spacing-space bin edges [R, S+1] when the producer has them as ONE tensor (the fused training path): what
    losses.ray_samples_to_sdist would otherwise cat together from spacing_starts / spacing_ends"""

    @property
    def shape(self):
        return self.frustums.shape

    def get_weights(self, densities: Tensor) -> Tensor:
        return get_weights(self, densities)

    def __getitem__(self, idx):
        """slicing along the sample axis, e.g. ray_samples[..., :-1] (models/neurad.py:388)."""
        # TensorDataclass semantics (utils/tensor_dataclass.py:119-148): the index addresses the BATCH dims,
        # the trailing feature dim of every field is kept.
        bidx = idx if isinstance(idx, tuple) else (idx,)

        def sl(t):
            return None if t is None else t[(*bidx, slice(None))] if Ellipsis in bidx else t[bidx]
        fr = self.frustums
        return replace(self, frustums=Frustums(sl(fr.origins), sl(fr.directions), sl(fr.starts), sl(fr.ends),
                                               sl(fr.pixel_area), sl(fr.offsets)),
                       camera_indices=sl(self.camera_indices), deltas=sl(self.deltas), spacing_starts=sl(self.spacing_starts),
                       spacing_ends=sl(self.spacing_ends), times=sl(self.times), sdist=None,
                       metadata=None if self.metadata is None else {k: sl(v) for k, v in self.metadata.items()})


@dataclass
class RayBundle:
    origins: Tensor       # [R,3]
    directions: Tensor    # [R,3]
    pixel_area: Tensor    # [R,1]
    camera_indices: Optional[Tensor] = None
    nears: Optional[Tensor] = None
    fars: Optional[Tensor] = None
    metadata: Dict[str, Tensor] = field(default_factory=dict)
    times: Optional[Tensor] = None
    termination_distances: Optional[Tensor] = None

    def __len__(self) -> int:
        return self.origins.numel() // self.origins.shape[-1]

    def get_ray_samples(self, bin_starts, bin_ends, spacing_starts=None, spacing_ends=None,
                        spacing_to_euclidean_fn=None) -> RaySamples:
        """rays.py:313-357: per-ray fields become [R,1,C] -> expanded (stride-0) views over the samples."""
        S = bin_starts.shape[-2]

        def ex(t):
            return None if t is None else t[..., None, :].expand(*t.shape[:-1], S, t.shape[-1])

        fr = Frustums(ex(self.origins), ex(self.directions), bin_starts, bin_ends, ex(self.pixel_area))
        return RaySamples(frustums=fr, camera_indices=ex(self.camera_indices), deltas=bin_ends - bin_starts,
                          spacing_starts=spacing_starts, spacing_ends=spacing_ends,
                          spacing_to_euclidean_fn=spacing_to_euclidean_fn,
                          metadata={k: ex(v) for k, v in self.metadata.items()}, times=ex(self.times))
