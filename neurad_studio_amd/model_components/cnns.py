"""The RGB decoder that follows the hot path for camera rays: rendered [patch, 48] features ->
3x-upsampled RGB patch (nerfstudio/models/neurad.py:198-216,359-366; model_components/cnns.py:20-46).

The modules hold the parameters (and run as they are on a CPU or for other decoder shapes); on a GPU ``decode_rgb`` runs
the reference-shaped decoder on the HIP kernels of csrc/decoder.hip.  Module nesting
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


def _fused_decoder_args(decoder: nn.Module):
    """the 38 parameters and 8 BatchNorm states of a make_rgb_decoder()-shaped module, or None when its shape is another
    one (then the torch modules run)"""
    try:
        first, blocks, up, last = decoder[0], [decoder[i] for i in (2, 3, 5, 6)], decoder[4], decoder[7]
    except (IndexError, TypeError):
        return None
    ok = (isinstance(first, nn.Conv2d) and first.kernel_size == (1, 1) and first.out_channels == 32 and first.in_channels <= 64
          and isinstance(up, nn.ConvTranspose2d) and up.kernel_size == (3, 3) and up.stride == (3, 3)
          and up.in_channels == up.out_channels == 32 and isinstance(last, nn.Conv2d) and last.kernel_size == (1, 1)
          and last.out_channels == 3 and last.in_channels == 32 and isinstance(decoder[8], nn.Sigmoid))
    params, states = [first.weight, first.bias], []
    from ..ops_decoder import _BnState

    for blk in blocks:
        mb = getattr(blk, "main_branch", None)
        if not ok or mb is None or len(mb) != 5 or not isinstance(getattr(blk, "res_branch", None), nn.Identity):
            return None
        ca, bna, cb, bnb = mb[0], mb[1], mb[3], mb[4]
        for cv, bn in ((ca, bna), (cb, bnb)):
            if not (isinstance(cv, nn.Conv2d) and cv.kernel_size == (7, 7) and cv.padding == (3, 3) and cv.in_channels == 32
                    and cv.out_channels == 32 and isinstance(bn, nn.BatchNorm2d) and bn.affine and bn.track_running_stats
                    and bn.momentum is not None):
                return None
            params += [cv.weight, cv.bias, bn.weight, bn.bias]
            states.append(_BnState(bn.running_mean, bn.running_var, bn.eps, bn.momentum))
    if not ok:
        return None
    params += [up.weight, up.bias, last.weight, last.bias]
    return params, states, [mb_bn for blk in blocks for mb_bn in (blk.main_branch[1], blk.main_branch[4])]


def decode_rgb(decoder: nn.Module, cam_features: Tensor, patch_size, fused: bool = True) -> Tensor:
    """decode_features' camera branch (models/neurad.py:361-366): [B*ph*pw, C] -> [B, ph*up, pw*up, 3].

    On a GPU, a decoder of the reference's shape runs on the HIP kernels of csrc/decoder.hip (ops_decoder.RgbDecoderFn: fp16
    operands, fp32 accumulation -- the reference trainer's mixed precision -- NHWC throughout, so neither permute happens);
    ``fused=False`` or any other decoder shape runs the torch modules."""
    if fused and cam_features.is_cuda and cam_features.dtype == torch.float32:
        args = _fused_decoder_args(decoder)
        if args is not None:
            from ..ops_decoder import RgbDecoderFn

            params, states, bns = args
            training = decoder.training
            rgb = RgbDecoderFn.apply(cam_features, tuple(patch_size), training, states, *params)
            if training:  # (one multi-tensor launch for the eight counters)
                torch._foreach_add_([bn.num_batches_tracked for bn in bns], 1)
            return rgb
    patches = cam_features.view(-1, *patch_size, cam_features.shape[-1]).permute(0, 3, 1, 2)
    return decoder(patches).permute(0, 2, 3, 1)
