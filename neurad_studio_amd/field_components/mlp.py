"""MLP with the reference's signature and ``layers.{k}.{weight,bias}`` state_dict layout
(nerfstudio/field_components/mlp.py:60-183), evaluated by the fp32 MFMA kernels."""
from __future__ import annotations

from typing import Optional

from torch import Tensor, nn

from .. import autograd as ag


class MLP(nn.Module):
    def __init__(self, in_dim: int, num_layers: int, layer_width: int, out_dim: Optional[int] = None,
                 skip_connections=None, activation: Optional[nn.Module] = nn.ReLU(),
                 out_activation: Optional[nn.Module] = None, implementation: str = "hip") -> None:
        super().__init__()
        if skip_connections:
            raise NotImplementedError("skip connections are not used on the NeuRAD hot path")
        if activation is not None and not isinstance(activation, nn.ReLU):
            raise NotImplementedError("the MFMA MLP kernel implements ReLU hidden activations (NeuRAD, neurad_field.py:103)")
        self.in_dim, self.num_layers, self.layer_width = in_dim, num_layers, layer_width
        self.out_dim = out_dim if out_dim is not None else layer_width
        self.activation, self.out_activation = activation, out_activation
        dims = [in_dim] + [layer_width] * (num_layers - 1) + [self.out_dim]
        self.layers = nn.ModuleList([nn.Linear(dims[k], dims[k + 1]) for k in range(num_layers)])

    def get_out_dim(self) -> int:
        return self.out_dim

    def forward(self, in_tensor: Tensor) -> Tensor:
        x = in_tensor.reshape(-1, self.in_dim)
        y = ag.mlp(x, [l.weight for l in self.layers], [l.bias for l in self.layers])
        if self.out_activation is not None:
            y = self.out_activation(y)
        return y.reshape(*in_tensor.shape[:-1], self.out_dim)
