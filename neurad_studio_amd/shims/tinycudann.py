"""tinycudann-shaped modules over the HIP kernels: ``Encoding``, ``Network``, ``NetworkWithInputEncoding`` with the
constructor / call forms neurad-studio uses (encodings.py:134-137,370-373,780-783; mlp.py:109-113,251-268).

Numerics follow the reference's *torch* branch, not tiny-cuda-nn (SURVEY §8a-H1'): every level hashed, floor()
scalings, fp32 parameters and outputs, ``nn.Linear`` layers WITH bias, SH evaluated on the [0,1]-normalised input as
given.  Put a directory with ``tinycudann/__init__.py: from neurad_studio_amd.shims.tinycudann import *`` on
PYTHONPATH and ``nerfstudio/utils/external.py:38-58`` picks it up with zero edits (run with use_4d_hashgrid=False)."""
from __future__ import annotations

import torch
from torch import Tensor, nn

from .. import autograd as ag
from .. import ops

_ACT = {"ReLU": nn.ReLU, "None": None, "Sigmoid": nn.Sigmoid}


class Encoding(nn.Module):
    def __init__(self, n_input_dims: int, encoding_config: dict, dtype=None, seed: int = 1337) -> None:
        super().__init__()
        self.n_input_dims, self.config = n_input_dims, dict(encoding_config)
        otype = encoding_config["otype"]
        if otype == "HashGrid":
            if n_input_dims != 3:
                raise NotImplementedError("4-D hash grids exist only in tiny-cuda-nn; use per-actor 3-D grids")
            if encoding_config.get("interpolation", "Linear") != "Linear":
                raise NotImplementedError("only Linear interpolation")
            L, F = encoding_config["n_levels"], encoding_config["n_features_per_level"]
            lg, base = encoding_config["log2_hashmap_size"], encoding_config["base_resolution"]
            growth = encoding_config.get("per_level_scale", 1.0)
            scalings = torch.floor(base * growth ** torch.arange(L)).to(torch.float32)  # encodings.py:350
            self.spec = ops.GridSpec(L, F, lg, base, int(scalings[-1].item()), scalings=scalings)
            g = torch.Generator().manual_seed(seed)
            self.params = nn.Parameter((torch.rand((L << lg) * F, generator=g) * 2 - 1) * 1e-3)
            self.n_output_dims = L * F
        elif otype == "SphericalHarmonics":
            if encoding_config.get("degree", 4) != 4:
                raise NotImplementedError("SH degree 4 only (neurad_field.py:108)")
            self.spec, self.n_output_dims = None, 16
            self.params = nn.Parameter(torch.zeros(0))
        else:
            raise NotImplementedError(f"encoding otype {otype} is not on the NeuRAD hot path")

    def forward(self, x: Tensor) -> Tensor:
        if self.spec is None:
            return ops.sh4_fwd(x.float().contiguous())
        table = self.params.view(self.spec.table_rows, self.spec.features_per_level)
        return ag.HashGridFn.apply(x.float().contiguous(), table, self.spec)


class Network(nn.Module):
    def __init__(self, n_input_dims: int, n_output_dims: int, network_config: dict, seed: int = 1337) -> None:
        super().__init__()
        if network_config.get("activation", "ReLU") != "ReLU":
            raise NotImplementedError("hidden activation must be ReLU")
        self.n_input_dims, self.n_output_dims = n_input_dims, n_output_dims
        width, nh = network_config["n_neurons"], network_config["n_hidden_layers"]
        dims = [n_input_dims] + [width] * nh + [n_output_dims]
        self.layers = nn.ModuleList([nn.Linear(dims[i], dims[i + 1]) for i in range(len(dims) - 1)])
        out_act = _ACT.get(network_config.get("output_activation", "None"))
        self.out_activation = out_act() if out_act is not None else None

    def forward(self, x: Tensor) -> Tensor:
        y = ag.mlp(x.float().contiguous(), [l.weight for l in self.layers], [l.bias for l in self.layers])
        return y if self.out_activation is None else self.out_activation(y)


class NetworkWithInputEncoding(nn.Module):
    def __init__(self, n_input_dims: int, n_output_dims: int, encoding_config: dict, network_config: dict,
                 seed: int = 1337) -> None:
        super().__init__()
        self.encoding = Encoding(n_input_dims, encoding_config, seed=seed)
        self.network = Network(self.encoding.n_output_dims, n_output_dims, network_config, seed=seed)
        self.n_input_dims, self.n_output_dims = n_input_dims, n_output_dims

    def forward(self, x: Tensor) -> Tensor:
        return self.network(self.encoding(x))
