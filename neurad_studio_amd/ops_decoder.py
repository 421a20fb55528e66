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


def conv7x7_pack_many(weights) -> Tensor:
    """up to 8 Conv2d weights -> [n, 2, ...] fragment-ordered fp16 weights (mode 0 and mode 1 of each) in ONE launch"""
    ws = [_chk(w.detach().to(torch.float32), "weight") for w in weights]
    for w in ws:
        if tuple(w.shape) != (32, 32, 7, 7):
            raise ValueError(f"conv7x7_pack_many: weight {tuple(w.shape)}, expected (32, 32, 7, 7)")
    out = torch.empty((len(ws), 2, _WFRAG_BYTES // 2), device=ws[0].device, dtype=torch.float16)
    ptrs = (C.c_void_p * len(ws))(*[w.data_ptr() for w in ws])
    call("nrhip_conv7x7_pack_many", ptrs, len(ws), _ptr(out), _stream())
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


def conv7x7_wgrad(x: Tensor, grad_out: Tensor, grad_weight: Tensor, grad_bias: Optional[Tensor] = None,
                  grad_scale: Optional[Tensor] = None) -> None:
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
    call("nrhip_conv7x7_wgrad", _ptr(x), _ptr(grad_out), _ptr(ws), _ptr(gw), _ptr(grad_bias), _ptr(grad_scale), b, h, w,
         _stream())


# ---- the other layers ------------------------------------------------------------------------------------------------
def _f32(t: Tensor, name: str) -> Tensor:
    return _chk(t.detach() if t.requires_grad else t, name)


def _ws(entry: str, *dims, device) -> Tensor:
    n = C.c_int64(0)
    call(entry, *dims, C.byref(n))
    return torch.empty((max(n.value, 1),), device=device, dtype=torch.float32)


def _act16(t: Tensor, name: str, like: Optional[Tensor] = None) -> Tensor:
    """an activation / gradient tensor of the decoder: contiguous cuda fp16 with 32 channels last (no CPU path)"""
    if not isinstance(t, Tensor) or not t.is_cuda or t.dtype != torch.float16 or not t.is_contiguous() or t.shape[-1] != 32:
        raise ValueError(f"{name}: expected a contiguous cuda fp16 tensor with 32 channels last, got "
                         f"{tuple(t.shape) if isinstance(t, Tensor) else type(t)} {getattr(t, 'dtype', '')} on {getattr(t, 'device', '?')}")
    if like is not None and t.shape != like.shape:
        raise ValueError(f"{name}: shape {tuple(t.shape)} differs from {tuple(like.shape)}")
    return t


def _coef(t: Tensor, name: str, rows: int) -> Tensor:
    if not isinstance(t, Tensor) or not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != rows * 32:
        raise ValueError(f"{name}: expected a contiguous cuda fp32 [{rows}, 32] tensor")
    return t


def bn_finalize(part: Tensor, count: int, gamma: Tensor, beta: Tensor, eps: float, momentum: float,
                running_mean: Optional[Tensor], running_var: Optional[Tensor]) -> Tensor:
    """-> coef [4, 32] = scale, shift, mean, rstd; running statistics updated in place like torch's batch_norm"""
    coef = torch.empty((4, 32), device=part.device, dtype=torch.float32)
    call("nrhip_dec_bn_finalize", _ptr(part), part.shape[0], count, _ptr(_f32(gamma, "gamma")), _ptr(_f32(beta, "beta")),
         float(eps), float(momentum), _ptr(running_mean), _ptr(running_var), _ptr(coef), _stream())
    return coef


def bn_act(c: Tensor, coef: Tensor, skip: Optional[Tensor] = None) -> Tensor:
    _act16(c, "c"), _coef(coef, "coef", 4)
    if skip is not None:
        _act16(skip, "skip", c)
    out = torch.empty_like(c)
    call("nrhip_dec_bn_act", _ptr(c), _ptr(coef), _ptr(skip), _ptr(out), c.numel() // 32, _stream())
    return out


def grad_scale(grad: Tensor) -> Tensor:
    """-> device [3] = {S, 1/S, scratch}: the power of two S that brings max |grad| into [0.5, 1) (the decoder backward's
    working scale: its fp16 gradient tensors carry S, every fp32 result is multiplied by 1/S)"""
    g = _chk(grad, "grad")
    scale = torch.empty((3,), device=g.device, dtype=torch.float32)
    call("nrhip_dec_grad_scale", _ptr(g), g.numel(), _ptr(scale), _stream())
    return scale


def bn_bwd(grad_out: Tensor, act: Tensor, c: Tensor, gamma: Tensor, coef: Tensor, grad_gamma: Tensor,
           grad_beta: Tensor, grad_scale: Optional[Tensor] = None) -> Tensor:
    _act16(c, "c"), _act16(grad_out, "grad_out", c), _act16(act, "act", c), _coef(coef, "coef", 4)
    npix = c.numel() // 32
    ws = _ws("nrhip_dec_bn_bwd_workspace", npix, device=c.device)
    out = torch.empty_like(c)
    call("nrhip_dec_bn_bwd", _ptr(grad_out), _ptr(act), _ptr(c), _ptr(_f32(gamma, "gamma")), _ptr(coef), _ptr(ws),
         _ptr(grad_gamma), _ptr(grad_beta), _ptr(grad_scale), _ptr(out), npix, _stream())
    return out


def add_masked(a: Tensor, grad_out: Tensor, act: Tensor) -> Tensor:
    _act16(a, "a"), _act16(grad_out, "grad_out", a), _act16(act, "act", a)
    out = torch.empty_like(a)
    call("nrhip_dec_add_masked", _ptr(a), _ptr(grad_out), _ptr(act), _ptr(out), a.numel() // 32, _stream())
    return out


def conv1x1_in_fwd(features: Tensor, weight: Tensor, bias: Tensor) -> Tensor:
    f = _chk(features, "features")
    n, cin = f.shape
    out = torch.empty((n, 32), device=f.device, dtype=torch.float16)
    call("nrhip_dec_conv1x1_in_fwd", _ptr(f), _ptr(_f32(weight, "weight").reshape(32, cin)), _ptr(_f32(bias, "bias")),
         _ptr(out), n, cin, _stream())
    return out


def conv1x1_in_bwd(features: Tensor, h: Tensor, grad_h: Tensor, weight: Tensor, grad_weight: Tensor, grad_bias: Tensor,
                   grad_scale: Optional[Tensor] = None):
    f = _chk(features, "features")
    n, cin = f.shape
    ws = _ws("nrhip_dec_conv1x1_in_bwd_workspace", n, cin, device=f.device)
    gf = torch.empty_like(f)
    call("nrhip_dec_conv1x1_in_bwd", _ptr(f), _ptr(h), _ptr(grad_h), _ptr(_f32(weight, "weight")), _ptr(ws), _ptr(gf),
         _ptr(grad_weight), _ptr(grad_bias), _ptr(grad_scale), n, cin, _stream())
    return gf


def upsample_pack(weight: Tensor) -> Tensor:
    """ConvTranspose2d weight [32, 32, 3, 3] -> fragment-ordered fp16 weights (forward, then input gradient)"""
    w = _f32(weight, "weight")
    if tuple(w.shape) != (32, 32, 3, 3):
        raise ValueError(f"upsample_pack: weight {tuple(w.shape)}, expected (32, 32, 3, 3)")
    out = torch.empty((2, 9 * 2 * 64 * 8), device=w.device, dtype=torch.float16)
    call("nrhip_dec_upsample_pack", _ptr(w), _ptr(out), _stream())
    return out


def upsample_fwd(h: Tensor, wup: Tensor, bias: Tensor) -> Tensor:
    _act16(h, "h")
    if h.dim() != 4 or wup.dtype != torch.float16 or wup.numel() != 2 * 9 * 2 * 64 * 8:
        raise ValueError("upsample_fwd: h must be [B, H, W, 32] and wup the output of upsample_pack")
    b, hh, w, _ = h.shape
    out = torch.empty((b, 3 * hh, 3 * w, 32), device=h.device, dtype=torch.float16)
    call("nrhip_dec_upsample_fwd", _ptr(h), _ptr(wup), _ptr(_f32(bias, "bias")), _ptr(out), b, hh, w, _stream())
    return out


def upsample_bwd(h: Tensor, grad_out: Tensor, wup: Tensor, grad_weight: Tensor, grad_bias: Tensor,
                 grad_scale: Optional[Tensor] = None) -> Tensor:
    _act16(h, "h"), _act16(grad_out, "grad_out")
    if h.dim() != 4 or grad_out.shape != (h.shape[0], 3 * h.shape[1], 3 * h.shape[2], 32) or wup.numel() != 2 * 9 * 2 * 64 * 8:
        raise ValueError("upsample_bwd: h [B, H, W, 32], grad_out [B, 3H, 3W, 32], wup from upsample_pack")
    b, hh, w, _ = h.shape
    ws = _ws("nrhip_dec_upsample_bwd_workspace", b, hh, w, device=h.device)
    gh = torch.empty_like(h)
    call("nrhip_dec_upsample_bwd", _ptr(h), _ptr(grad_out), _ptr(wup), _ptr(ws), _ptr(gh), _ptr(grad_weight),
         _ptr(grad_bias), _ptr(grad_scale), b, hh, w, _stream())
    return gh


def rgb_fwd(h: Tensor, weight: Tensor, bias: Tensor) -> Tensor:
    _act16(h, "h")
    if h.dim() != 4 or weight.numel() != 96 or bias.numel() != 3:
        raise ValueError("rgb_fwd: h must be [B, H, W, 32], weight [3, 32(, 1, 1)], bias [3]")
    b, hh, w, _ = h.shape
    rgb = torch.empty((b, hh, w, 3), device=h.device, dtype=torch.float32)
    call("nrhip_dec_rgb_fwd", _ptr(h), _ptr(_f32(weight, "weight")), _ptr(_f32(bias, "bias")), _ptr(rgb), b * hh * w,
         _stream())
    return rgb


def rgb_bwd(h: Tensor, rgb: Tensor, grad_rgb: Tensor, weight: Tensor, grad_weight: Tensor, grad_bias: Tensor,
            grad_scale: Optional[Tensor] = None) -> Tensor:
    _act16(h, "h")
    npix = h.numel() // 32
    if rgb.numel() != 3 * npix or grad_rgb.numel() != 3 * npix or grad_weight.numel() != 96 or grad_bias.numel() != 3:
        raise ValueError("rgb_bwd: rgb / grad_rgb must hold 3 values per pixel of h, grad_weight 96, grad_bias 3")
    ws = _ws("nrhip_dec_rgb_bwd_workspace", npix, device=h.device)
    gh = torch.empty_like(h)
    call("nrhip_dec_rgb_bwd", _ptr(h), _ptr(rgb), _ptr(_chk(grad_rgb, "grad_rgb")), _ptr(_f32(weight, "weight")), _ptr(ws),
         _ptr(gh), _ptr(grad_weight), _ptr(grad_bias), _ptr(grad_scale), npix, _stream())
    return gh


# ---- the decoder as one autograd node --------------------------------------------------------------------------------
class _BnState:
    """what BatchNorm2d carries besides its parameters (not differentiable)"""

    def __init__(self, running_mean, running_var, eps, momentum):
        self.running_mean, self.running_var, self.eps, self.momentum = running_mean, running_var, eps, momentum


def _decoder_struct(b: int, ph: int, pw: int, cin: int, training: bool, bn_states, p):
    """-> (nrhip_rgb_decoder, keep-alive list): params p = w0, b0, 4 x (wa, ba, gamma1, beta1, wb, bb, gamma2, beta2), wu, bu,
    wo, bo in torch layouts; contiguous fp32 copies are made where a parameter is not one already"""
    from ._lib import RgbDecoder

    keep = [_f32(t, "decoder parameter") for t in p]
    d = RgbDecoder()
    d.n_patches, d.patch_h, d.patch_w, d.cin, d.training = b, ph, pw, cin, int(training)
    d.conv_in_w, d.conv_in_b = keep[0].data_ptr(), keep[1].data_ptr()
    for i in range(8):
        w, bias, gamma, beta = keep[2 + 4 * i:6 + 4 * i]
        if tuple(w.shape) != (32, 32, 7, 7):
            raise ValueError(f"decode_rgb: convolution weight {tuple(w.shape)}, expected (32, 32, 7, 7)")
        d.conv_w[i], d.conv_b[i], d.bn_gamma[i], d.bn_beta[i] = w.data_ptr(), bias.data_ptr(), gamma.data_ptr(), beta.data_ptr()
        st = bn_states[i]
        for t in (st.running_mean, st.running_var):
            if t.dtype != torch.float32 or not t.is_contiguous() or not t.is_cuda:
                raise ValueError("decode_rgb: BatchNorm running statistics must be contiguous cuda fp32 tensors")
        d.bn_running_mean[i], d.bn_running_var[i] = st.running_mean.data_ptr(), st.running_var.data_ptr()
        d.bn_eps[i], d.bn_momentum[i] = float(st.eps), float(st.momentum)
    d.up_w, d.up_b, d.out_w, d.out_b = (t.data_ptr() for t in keep[34:38])
    return d, keep


class RgbDecoderFn(torch.autograd.Function):
    """rgb_decoder of models/neurad.py:198-216 on [n, cin] feature rows of ph x pw patches -> rgb [B, 3 ph, 3 pw, 3] fp32: ONE
    call into the library per direction (nrhip_rgb_decoder_fwd / _bwd).
    params: w0, b0, 4 x (wa, ba, gamma1, beta1, wb, bb, gamma2, beta2), wu, bu, wo, bo (38 tensors, torch layouts)."""

    @staticmethod
    def forward(ctx, features, patch, training, bn_states, *params):
        ph, pw = patch
        n = features.shape[0]
        if n == 0 or n % (ph * pw):
            raise ValueError(f"decode_rgb: {n} feature rows are not whole {ph}x{pw} patches")
        b = n // (ph * pw)
        feats = _chk(features.detach(), "features")
        d, keep = _decoder_struct(b, ph, pw, feats.shape[1], training, bn_states, params)
        sizes = [C.c_int64(0), C.c_int64(0), C.c_int64(0)]
        call("nrhip_rgb_decoder_sizes", C.byref(d), *(C.byref(v) for v in sizes))
        saved = torch.empty((sizes[0].value,), device=feats.device, dtype=torch.uint8)
        work = torch.empty((sizes[1].value,), device=feats.device, dtype=torch.uint8)
        rgb = torch.empty((b, 3 * ph, 3 * pw, 3), device=feats.device, dtype=torch.float32)
        call("nrhip_rgb_decoder_fwd", C.byref(d), _ptr(feats), _ptr(saved), _ptr(work), _ptr(rgb), _stream())
        ctx.training, ctx.dec, ctx.keep = training, d, keep
        # through save_for_backward, not as attributes: `rgb` is this node's own output (an attribute would close the cycle
        # rgb -> grad_fn -> ctx -> rgb that only the cyclic GC breaks, holding the activation buffer until then), the
        # buffers are released when backward has run, and in-place edits of them are detected
        ctx.save_for_backward(feats, saved, rgb)
        ctx.sizes = sizes
        ctx.shapes = [t.shape for t in params]
        ctx.need = [features.requires_grad] + [t.requires_grad for t in params]
        return rgb

    @staticmethod
    def backward(ctx, grad_rgb):
        feats, saved, rgb = ctx.saved_tensors
        grad_rgb = _chk(grad_rgb.contiguous().float(), "grad_rgb")
        work = torch.empty((ctx.sizes[1].value,), device=feats.device, dtype=torch.uint8)
        gf = torch.empty_like(feats)
        flat = torch.empty((ctx.sizes[2].value,), device=feats.device, dtype=torch.float32)
        call("nrhip_rgb_decoder_bwd", C.byref(ctx.dec), _ptr(feats), _ptr(saved), _ptr(rgb), _ptr(grad_rgb), _ptr(work),
             _ptr(gf), _ptr(flat), _stream())
        grads, off = [], 0
        for shape, need in zip(ctx.shapes, ctx.need[1:]):
            k = 1
            for v in shape:
                k *= v
            grads.append(flat[off:off + k].view(shape) if need else None)
            off += k
        return (gf if ctx.need[0] else None, None, None, None, *grads)
