"""NeuRADHashEncoding: static-world hash grid with ZipNeRF-style down-weighting
(mirror of nerfstudio/field_components/neurad_encoding.py:34-304, static path).

The contraction (H3), gaussian (H2), lookup (H1) and rescale (H4) are ONE HIP kernel (nrhip_encode_fwd) that
consumes per-ray origin/direction and per-sample [start,end]; nothing of shape [R,S,3] is materialised.
Dynamic actors (H5) are the next row of SURVEY §8: a non-empty ``dynamic_actors`` raises."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, Optional, Tuple

import torch
from torch import Tensor, nn

from .. import autograd as ag
from .encodings import HashEncoding


@dataclass
class StaticSettings:  # neurad_encoding.py:34-45
    hashgrid_dim: int = 4
    num_levels: int = 8
    base_res: int = 32
    max_res: int = 8192
    log2_hashmap_size: int = 22


@dataclass
class ActorSettings:  # neurad_encoding.py:48-66
    flip_prob: float = 0.5
    actor_scale: float = 10.0
    hashgrid_dim: int = 4
    num_levels: int = 4
    base_res: int = 64
    max_res: int = 1024
    log2_hashmap_size: int = 17
    use_4d_hashgrid: bool = False  # the 4-D grid exists only in tiny-cuda-nn (SURVEY §8b caveat)


@dataclass
class NeuRADHashEncodingConfig:  # neurad_encoding.py:69-82
    static: StaticSettings = field(default_factory=StaticSettings)
    actor: ActorSettings = field(default_factory=ActorSettings)
    disable_actors: bool = False
    require_actor_grad: bool = True

    def setup(self, **kwargs):
        return NeuRADHashEncoding(self, **kwargs)


class NeuRADHashEncoding(nn.Module):
    def __init__(self, config: NeuRADHashEncodingConfig, dynamic_actors=None, static_scale: float = 1.0,
                 implementation: str = "hip") -> None:
        super().__init__()
        self.config, self.implementation, self.actors = config, implementation, dynamic_actors
        n_actors = 0 if dynamic_actors is None else int(getattr(dynamic_actors, "n_actors", 0))
        if n_actors > 0 and not config.disable_actors:
            raise NotImplementedError("dynamic actors (SURVEY §8a-H5) are not part of this round's HIP path")
        self.static_scale = float(static_scale)
        s = config.static
        self.static_grid = HashEncoding(num_levels=s.num_levels, min_res=s.base_res, max_res=s.max_res,
                                        log2_hashmap_size=s.log2_hashmap_size, features_per_level=s.hashgrid_dim,
                                        implementation=implementation)
        self.actor_grids = nn.ModuleList([])
        self.scene_repr_dim = self.static_grid.get_out_dim()

    def get_out_dim(self) -> int:
        return self.scene_repr_dim

    def get_param_groups(self, param_groups: Dict):
        param_groups["hashgrids"] += list(self.static_grid.parameters()) + list(self.actor_grids.parameters())

    def forward_rays(self, origins, directions, pixel_area, starts, ends) -> Tensor:
        """-> [R*S, L*F] rescaled static features (autograd: table gradient via scatter-add atomics)."""
        g = self.static_grid
        return ag.EncodeFn.apply(g.hash_table, g.spec, self.static_scale, origins, directions, pixel_area, starts, ends)

    def forward(self, ray_samples, times=None, directions: Optional[Tensor] = None) -> Tuple[Tensor, Optional[Tensor]]:
        """Reference signature is forward(GaussiansStd, times, directions); here the frustums are passed directly
        (the gaussian is computed inside the kernel).  Returns (features [N, L*F], directions)."""
        fr = ray_samples.frustums
        o, d, a = fr.per_ray()
        feats = self.forward_rays(o, d, a, fr.starts[..., 0], fr.ends[..., 0])
        return feats, directions
