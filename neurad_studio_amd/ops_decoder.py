"""ctypes wrappers of csrc/decoder.hip: the RGB CNN decoder's kernels.  Activations are NHWC fp16
tensors [B, H, W, 32]; no CPU path."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch
from torch import Tensor

from ._lib import call
from .ops import _chk, _ptr, _stream

_WFRAG_BYTES = 49 * 2 * 64 * 16


def conv7x7_pack(weight: Tensor, mode: int = 0) -> Tensor:
    """torch Conv2d weight [32,32,7,7] fp32 -> fragment-ordered fp16 weights.  mode 0: forward; 1: input gradient."""
    w = _chk(weight.detach().to(torch.float32), "weight")
    if tuple(w.shape) != (32, 32, 7, 7):
        raise ValueError(f"conv7x7_pack: weight {tuple(w.shape)}, expected (32, 32, 7, 7)")
    out = torch.empty((_WFRAG_BYTES // 2,), device=w.device, dtype=torch.float16)
    call("nrhip_conv7x7_pack", _ptr(w), mode, _ptr(out), _stream())
    return out


def _rows_per_wave(h: int, w: int, b: int) -> int:
    # 16 x 32 output tiles once they fill the chip, smaller ones for the 32 x 32 stage (40 patches = 80 tiles of 16 rows)
    for r in (4, 2, 1):
        if b * ((w + 31) // 32) * ((h + 4 * r - 1) // (4 * r)) >= 512:
            return r
    return 1


def conv7x7(x: Tensor, wfrag: Tensor, bias: Optional[Tensor] = None, stats: bool = False,
            rows_per_wave: Optional[int] = None) -> Tuple[Tensor, Optional[Tensor]]:
    """x [B,H,W,32] fp16 -> (conv(x) + bias as [B,H,W,32] fp16, per-workgroup [n, 64] sums / sums of squares | None)"""
    if x.dtype != torch.float16 or x.dim() != 4 or x.shape[-1] != 32 or not x.is_contiguous() or not x.is_cuda:
        raise ValueError("conv7x7: x must be a contiguous cuda fp16 [B, H, W, 32] tensor")
    b, h, w, _ = x.shape
    r = rows_per_wave or _rows_per_wave(h, w, b)
    out = torch.empty_like(x)
    part = None
    if stats:
        tiles = C.c_int32(0)
        call("nrhip_conv7x7_tiles", h, w, r, C.byref(tiles))
        part = torch.empty((b * tiles.value, 64), device=x.device, dtype=torch.float32)
    bias_f = None if bias is None else _chk(bias.detach().to(torch.float32), "bias")
    call("nrhip_conv7x7", _ptr(x), _ptr(wfrag), _ptr(bias_f), _ptr(out), _ptr(part), b, h, w, r, _stream())
    return out, part


def conv7x7_wgrad(x: Tensor, grad_out: Tensor, grad_weight: Tensor, grad_bias: Optional[Tensor] = None) -> None:
    """grad_weight [32,32,7,7] (+ grad_bias [32]) += the convolution's weight (bias) gradient; x, grad_out NHWC fp16"""
    for t, n in ((x, "x"), (grad_out, "grad_out")):
        if t.dtype != torch.float16 or t.dim() != 4 or t.shape[-1] != 32 or not t.is_contiguous() or not t.is_cuda:
            raise ValueError(f"conv7x7_wgrad: {n} must be a contiguous cuda fp16 [B, H, W, 32] tensor")
    if x.shape != grad_out.shape:
        raise ValueError("conv7x7_wgrad: x and grad_out differ in shape")
    b, h, w, _ = x.shape
    gw = _chk(grad_weight, "grad_weight")
    if gw.data_ptr() != grad_weight.data_ptr() or tuple(gw.shape) != (32, 32, 7, 7):
        raise ValueError("conv7x7_wgrad: grad_weight must be a contiguous fp32 [32, 32, 7, 7] tensor")
    n = C.c_int64(0)
    call("nrhip_conv7x7_wgrad_workspace", b, h, w, C.byref(n))
    ws = torch.empty((n.value,), device=x.device, dtype=torch.float32)
    call("nrhip_conv7x7_wgrad", _ptr(x), _ptr(grad_out), _ptr(ws), _ptr(gw), _ptr(grad_bias), b, h, w, _stream())
