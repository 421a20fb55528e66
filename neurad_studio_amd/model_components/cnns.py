"""The RGB decoder that follows the hot path for camera rays: rendered [patch, 48] features ->
3x-upsampled RGB patch (nerfstudio/models/neurad.py:198-216,359-366; model_components/cnns.py:20-46).

Plain torch modules: on ROCm the convolutions are MIOpen's (the survey's "MIOpen first, fuse later").  Module nesting
and indices reproduce the reference's state_dict names (``rgb_decoder.{0,2,3,4,5,6,7}``,
``....main_branch.{0,1,3,4}``), so a neurad checkpoint's decoder loads as is."""
from __future__ import annotations

import torch
from torch import Tensor, nn


class BasicBlock(nn.Module):
    """conv-BN-ReLU-conv-BN with an identity skip, ReLU after the sum (cnns.py:20-46, in_dim == dim)"""

    def __init__(self, dim: int, kernel_size: int = 7, use_bn: bool = True) -> None:
        super().__init__()
        pad = kernel_size // 2
        norm = (lambda: nn.BatchNorm2d(dim)) if use_bn else nn.Identity
        self.res_branch = nn.Identity()
        self.main_branch = nn.Sequential(nn.Conv2d(dim, dim, kernel_size, padding=pad), norm(), nn.ReLU(inplace=True),
                                         nn.Conv2d(dim, dim, kernel_size, padding=pad), norm())
        self.final_activation = nn.ReLU(inplace=True)

    def forward(self, x: Tensor) -> Tensor:
        return self.final_activation(self.res_branch(x) + self.main_branch(x))


def make_rgb_decoder(in_dim: int = 48, hidden_dim: int = 32, upsample: int = 3) -> nn.Sequential:
    return nn.Sequential(
        nn.Conv2d(in_dim, hidden_dim, kernel_size=1), nn.ReLU(inplace=True),
        BasicBlock(hidden_dim), BasicBlock(hidden_dim),
        nn.ConvTranspose2d(hidden_dim, hidden_dim, kernel_size=upsample, stride=upsample),
        BasicBlock(hidden_dim), BasicBlock(hidden_dim),
        nn.Conv2d(hidden_dim, 3, kernel_size=1), nn.Sigmoid())


def decode_rgb(decoder: nn.Module, cam_features: Tensor, patch_size) -> Tensor:
    """decode_features' camera branch (models/neurad.py:361-366): [B*ph*pw, C] -> [B, ph*up, pw*up, 3]"""
    patches = cam_features.view(-1, *patch_size, cam_features.shape[-1]).permute(0, 3, 1, 2)
    return decoder(patches).permute(0, 2, 3, 1)
