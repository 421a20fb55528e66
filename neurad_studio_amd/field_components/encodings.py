"""HashEncoding / SHEncoding with the reference's constructor signature and state_dict layout
(nerfstudio/field_components/encodings.py:311-471,760-805), backed by the HIP kernels.

``implementation`` accepts "hip" (and, for drop-in configs, "tcnn"/"torch" which all select the HIP path --
numerics follow the reference's *torch* branch: every level hashed, floor() scalings, fp32)."""
from __future__ import annotations

from typing import Optional

import torch
from torch import Tensor, nn

from .. import autograd as ag
from .. import ops


class HashEncoding(nn.Module):
    def __init__(self, num_levels: int = 16, min_res: int = 16, max_res: int = 1024, log2_hashmap_size: int = 19,
                 features_per_level: int = 2, hash_init_scale: float = 0.001, implementation: str = "hip",
                 interpolation: Optional[str] = None, n_input_dims: int = 3) -> None:
        super().__init__()
        if n_input_dims != 3:
            raise NotImplementedError("4-D actor hash grid exists only in tiny-cuda-nn (no torch oracle, SURVEY §8c); "
                                      "use per-actor 3-D grids (use_4d_hashgrid=False)")
        assert interpolation is None or interpolation == "Linear", f"interpolation '{interpolation}' is not supported"
        self.in_dim = 3
        self.num_levels, self.min_res, self.max_res = num_levels, min_res, max_res
        self.features_per_level, self.log2_hashmap_size = features_per_level, log2_hashmap_size
        self.hash_table_size = 2**log2_hashmap_size
        self.hash_init_scale = hash_init_scale
        self.spec = ops.GridSpec(num_levels, features_per_level, log2_hashmap_size, min_res, max_res)
        self.register_buffer("scalings", self.spec.scalings.clone())
        table = torch.rand(size=(self.hash_table_size * num_levels, features_per_level)) * 2 - 1  # encodings.py:382-384
        self.hash_table = nn.Parameter(table * hash_init_scale)

    def get_out_dim(self) -> int:
        return self.num_levels * self.features_per_level

    def forward(self, in_tensor: Tensor) -> Tensor:
        assert in_tensor.shape[-1] == 3
        flat = in_tensor.reshape(-1, 3)
        out = ag.HashGridFn.apply(flat, self.hash_table, self.spec)
        return out.reshape(*in_tensor.shape[:-1], self.get_out_dim())


class SHEncoding(nn.Module):
    def __init__(self, levels: int = 4, implementation: str = "hip") -> None:
        super().__init__()
        if levels != 4:
            raise ValueError("the HIP SH kernel is instantiated for levels=4 (NeuRAD, neurad_field.py:108)")
        self.levels, self.in_dim = levels, 3

    def get_out_dim(self) -> int:
        return self.levels**2

    @torch.no_grad()
    def forward(self, in_tensor: Tensor) -> Tensor:
        return ops.sh4_fwd(in_tensor.reshape(-1, 3).float()).reshape(*in_tensor.shape[:-1], 16)
