"""Tensor-level wrappers over the C ABI (no autograd here; see autograd.py).

PyTorch is plumbing only: it owns the device buffers and the current HIP stream.  Every function
launches hand-written HIP kernels from libneurad_hip.so on ``torch.cuda.current_stream()`` and fails
loudly when given CPU tensors -- there is no fallback path.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch
from torch import Tensor

from . import _lib
from ._lib import call


def _stream() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t: Optional[Tensor]) -> C.c_void_p:
    return C.c_void_p(0 if t is None else t.data_ptr())


def _chk(t: Tensor, name: str, dtype=torch.float32) -> Tensor:
    if not isinstance(t, Tensor):
        raise TypeError(f"{name}: expected a torch.Tensor, got {type(t)}")
    if not t.is_cuda:
        raise _lib.NeuradHipError(f"{name}: tensor is on {t.device}; the HIP path needs a GPU tensor (no CPU fallback)")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


def hash_scalings(num_levels: int, min_res: int, max_res: int) -> Tensor:
    """scalings_l = floor(min_res * g**l) evaluated exactly like encodings.py:347-350 (fp32 torch ops on CPU)."""
    levels = torch.arange(num_levels)
    growth = np.exp((np.log(max_res) - np.log(min_res)) / (num_levels - 1)) if num_levels > 1 else 1.0
    return torch.floor(min_res * growth**levels).to(torch.float32)


@dataclass
class GridSpec:
    """Static description of one HashEncoding (encodings.py:326-352)."""

    num_levels: int
    features_per_level: int
    log2_hashmap_size: int
    min_res: int
    max_res: int
    scalings: Optional[Tensor] = None  # CPU fp32 [L]

    def __post_init__(self):
        if self.scalings is None:
            self.scalings = hash_scalings(self.num_levels, self.min_res, self.max_res)

    @property
    def out_dim(self) -> int:
        return self.num_levels * self.features_per_level

    @property
    def table_rows(self) -> int:
        return self.num_levels << self.log2_hashmap_size

    def c_grid(self, table: Tensor) -> _lib.Grid:
        if table.dtype not in (torch.float32, torch.float16):
            raise TypeError(f"hash table must be fp32 or fp16, got {table.dtype}")
        if tuple(table.shape) != (self.table_rows, self.features_per_level):
            raise ValueError(f"hash table shape {tuple(table.shape)} != {(self.table_rows, self.features_per_level)}")
        g = _lib.Grid()
        g.num_levels, g.n_features = self.num_levels, self.features_per_level
        g.log2_table_size = self.log2_hashmap_size
        g.param_dtype = 1 if table.dtype == torch.float16 else 0
        sc = self.scalings.tolist()
        for i, v in enumerate(sc):
            g.scalings[i] = v
        return g


def _c_mlp(weights: Sequence[Tensor], biases: Sequence[Optional[Tensor]]) -> Tuple[_lib.Mlp, list]:
    n = len(weights)
    if not 1 <= n <= _lib.MAX_LAYERS:
        raise ValueError(f"MLP needs 1..{_lib.MAX_LAYERS} layers, got {n}")
    keep = []
    m = _lib.Mlp()
    m.num_layers = n
    m.in_dim = weights[0].shape[1]
    m.out_dim = weights[-1].shape[0]
    m.hidden_dim = weights[0].shape[0] if n > 1 else 0
    for k, (w, b) in enumerate(zip(weights, biases)):
        exp_in = m.in_dim if k == 0 else m.hidden_dim
        exp_out = m.out_dim if k == n - 1 else m.hidden_dim
        if tuple(w.shape) != (exp_out, exp_in):
            raise ValueError(f"layer {k}: weight shape {tuple(w.shape)} != {(exp_out, exp_in)} (uniform hidden width)")
        w = _chk(w, f"weight[{k}]")
        keep.append(w)
        m.weight[k] = w.data_ptr()
        if b is not None:
            b = _chk(b, f"bias[{k}]")
            keep.append(b)
            m.bias[k] = b.data_ptr()
        else:
            m.bias[k] = None
    return m, keep


def _c_rays(origins: Tensor, directions: Tensor, pixel_area: Tensor, starts: Tensor, ends: Tensor,
            order: Optional[Tensor] = None):
    """starts/ends: [R,S] each, or views into one [R,S+1] edge tensor (stride S+1) -- no copies are made.
    order: optional int32 [R] processing order from ``ray_order`` (locality hint for the fused kernels)."""
    o = _chk(origins, "origins")
    d = _chk(directions, "directions")
    a = _chk(pixel_area.reshape(-1), "pixel_area")
    R = o.shape[0]
    if starts.dim() != 2 or starts.shape != ends.shape or starts.shape[0] != R:
        raise ValueError(f"starts/ends must be [R,S] with R={R}; got {tuple(starts.shape)}, {tuple(ends.shape)}")
    if not (starts.is_cuda and ends.is_cuda and starts.dtype == torch.float32 and ends.dtype == torch.float32):
        raise _lib.NeuradHipError("starts/ends must be fp32 GPU tensors")
    S = starts.shape[1]
    if starts.stride(1) != 1 or ends.stride(1) != 1 or starts.stride(0) != ends.stride(0):
        starts, ends = starts.contiguous(), ends.contiguous()
    r = _lib.Rays()
    r.n_rays, r.n_samples = R, S
    r.origins, r.directions, r.pixel_area = o.data_ptr(), d.data_ptr(), a.data_ptr()
    r.starts, r.ends = starts.data_ptr(), ends.data_ptr()
    r.sample_stride = starts.stride(0) if R > 1 else max(S, 1)
    if order is not None:
        order = _chk(order, "order", torch.int32)
        if order.shape != (R,):
            raise ValueError(f"order must be int32 [R={R}], got {tuple(order.shape)}")
        r.order = order.data_ptr()
    return r, (o, d, a, starts, ends, order)


_RAY_ORDER_LARGE = 16384  # rays above which ray_order takes the multi-workgroup counting sort


def ray_order(origins: Tensor, directions: Tensor, static_scale: float, t_ref: Optional[float] = None,
              key_bits: int = 0) -> Tensor:
    """Processing order that groups rays looking at the same region (csrc/rayorder.hip) -> int32 [R] permutation to pass
    as ``order=`` to field_fwd / field_fwd_train / render_fwd.  t_ref: distance of the key point along the ray; default
    = static_scale, the contraction boundary, where the key cell is set by where the ray leaves the scene."""
    t_ref = float(static_scale) if t_ref is None else t_ref
    o, d = _chk(origins, "origins"), _chk(directions, "directions")
    out = torch.empty((o.shape[0],), device=o.device, dtype=torch.int32)
    if o.shape[0] > _RAY_ORDER_LARGE:  # many workgroups: the one-workgroup pass costs ~2 us per 1024 rays
        need = C.c_int64(0)
        call("nrhip_ray_order_workspace", o.shape[0], int(key_bits), C.byref(need))
        ws = torch.empty((need.value,), device=o.device, dtype=torch.uint8)
        call("nrhip_ray_order_large", _ptr(o), _ptr(d), o.shape[0], float(t_ref), float(static_scale), int(key_bits),
             _ptr(ws), ws.numel(), _ptr(out), _stream())
        return out
    call("nrhip_ray_order", _ptr(o), _ptr(d), o.shape[0], float(t_ref), float(static_scale), int(key_bits), _ptr(out),
         _stream())
    return out


# ------------------------------------------------------------------------------------------------
def hashgrid_fwd(spec: GridSpec, table: Tensor, x: Tensor) -> Tensor:
    x = _chk(x, "x")
    if x.dim() != 2 or x.shape[1] != 3:
        raise ValueError(f"x must be [N,3], got {tuple(x.shape)}")  # encodings.py:428
    table = table if table.is_contiguous() else table.contiguous()
    out = torch.empty((x.shape[0], spec.out_dim), device=x.device, dtype=torch.float32)
    g = spec.c_grid(table)
    call("nrhip_hashgrid_fwd", C.byref(g), _ptr(table), _ptr(x), x.shape[0], _ptr(out), _stream())
    return out


def hashgrid_bwd(spec: GridSpec, table_like: Tensor, x: Tensor, grad_out: Tensor) -> Tensor:
    x, grad_out = _chk(x, "x"), _chk(grad_out, "grad_out")
    gt = torch.empty((spec.table_rows, spec.features_per_level), device=x.device, dtype=torch.float32)
    g = spec.c_grid(gt)
    ws = _table_grad_workspace(g, x.shape[0], x.device)
    if ws is not None:  # overwrite = 1: the partition writes every element of the gradient, no zero-fill
        call("nrhip_hashgrid_bwd_binned", C.byref(g), _ptr(x), _ptr(grad_out), x.shape[0], _ptr(gt), 1, _ptr(ws),
             ws.numel(), _stream())
    else:
        call("nrhip_hashgrid_bwd", C.byref(g), _ptr(x), _ptr(grad_out), x.shape[0], _ptr(gt.zero_()), _stream())
    return gt


_POINTER_TABLES: dict = {}


def _ptr_array(tensors: Sequence[Tensor]) -> Tensor:
    """device array of data pointers (the multi-grid entry points take `void* const*` in device memory); uploaded once
    per distinct set of tensors -- a pageable H2D copy per call synchronises the host with the stream"""
    key = (tensors[0].device, tuple(t.data_ptr() for t in tensors))
    ptrs = _POINTER_TABLES.get(key)
    if ptrs is None:
        if len(_POINTER_TABLES) >= 32:
            _POINTER_TABLES.clear()
        ptrs = _POINTER_TABLES[key] = torch.tensor(key[1], dtype=torch.int64, device=key[0])
    return ptrs


def hashgrid_multi_fwd(spec: GridSpec, tables: Sequence[Tensor], grid_id: Tensor, x: Tensor) -> Tensor:
    """sample i -> tables[grid_id[i]]: all actor grids in one launch"""
    x, grid_id = _chk(x, "x"), _chk(grid_id, "grid_id", torch.int32)
    tables = [_chk(t, "table", tables[0].dtype) for t in tables]  # fp32 or fp16 storage, one dtype per call
    out = torch.empty((x.shape[0], spec.out_dim), device=x.device, dtype=torch.float32)
    g = spec.c_grid(tables[0])
    ptrs = _ptr_array(tables)
    call("nrhip_hashgrid_multi_fwd", C.byref(g), _ptr(ptrs), len(tables), _ptr(grid_id), _ptr(x), x.shape[0], _ptr(out),
         _stream())
    return out


def grids_present(grid_id: Tensor, n_grids: int) -> List[bool]:
    """which grids a batch of rows refers to, as a host list (ONE device->host read; callers do it in the forward, right
    behind the read that sized the batch, so that the backward needs none)"""
    # (a scatter of ones, not torch.bincount: bincount first reduces max(grid_id) over all rows -- 0.25 ms at 0.5 M rows)
    return torch.zeros(n_grids, device=grid_id.device, dtype=torch.uint8).index_fill_(0, grid_id.long(), 1).tolist()


def hashgrid_multi_bwd(spec: GridSpec, n_grids: int, grid_id: Tensor, x: Tensor, grad_out: Tensor,
                       present: Optional[List[bool]] = None, out_dtype=torch.float32, dense_block: bool = False):
    """-> one gradient per grid, None for grids no sample refers to (like the reference's per-id loop, which never
    touches them: their optimizer state must not decay).  present: ``grids_present(grid_id, n_grids)`` when the caller
    already has it -- the backward then runs without a device->host read and without a host->device copy.
    dense_block=True: -> ONE tensor [n_grids, rows, F] instead (zeros for the grids without samples)."""
    x, grid_id, grad_out = _chk(x, "x"), _chk(grid_id, "grid_id", torch.int32), _chk(grad_out, "grad_out")
    if present is None:
        present = grids_present(grid_id, n_grids)
    present = [bool(p) for p in present]
    if dense_block:
        # slot a = grid a; a grid without samples keeps its (all-zero) slot
        block = _multi_bwd_block(spec, n_grids, grid_id, x, grad_out, tuple(a if p else -1 for a, p in enumerate(present)), n_grids,
                                 out_dtype) if any(present) else None
        if block is None:
            block = torch.zeros((n_grids, spec.table_rows, spec.features_per_level), device=x.device, dtype=out_dtype)
        return block
    if not any(present):
        return [None] * n_grids
    slots, k = [], 0
    for p in present:
        slots.append(k if p else -1)
        k += int(p)
    views = iter(_multi_bwd_block(spec, n_grids, grid_id, x, grad_out, tuple(slots), k, out_dtype).unbind(0))
    return [next(views) if p else None for p in present]


def _multi_bwd_block(spec: GridSpec, n_grids: int, grid_id: Tensor, x: Tensor, grad_out: Tensor, slots: Tuple[int, ...],
                     n_slots: int, out_dtype) -> Tensor:
    """-> [n_slots, rows, F] of ``out_dtype``: slot slots[a] holds grid a's gradient (slots[a] < 0: grid a sends nothing)"""
    n = x.shape[0]
    if n >= _BINNED_MIN_SAMPLES and not _FORCE_ATOMIC_SCATTER and _MULTI_BWD_BINNED:
        # the radix partition over (slot, level, slice) (csrc/encode_bwd_binned.hip, MultiSrc): no memory-side atomics, every
        # element of the block written by the partition (no zero-fill), fp16-storage grids get their fp16 gradient directly
        half = out_dtype == torch.float16 and n <= _BINNED_ROUND_SAMPLES and "NRHIP_BIN_ROUND_LOG2" not in os.environ
        g = spec.c_grid(torch.empty((spec.table_rows, spec.features_per_level), device="meta"))  # (the shape only)
        need = C.c_int64(0)
        call("nrhip_hashgrid_multi_bwd_binned_workspace", C.byref(g), n_slots, n, C.byref(need))
        if need.value > 0:
            block = torch.empty((n_slots, spec.table_rows, spec.features_per_level), device=x.device,
                                dtype=torch.float16 if half else torch.float32)
            ws = torch.empty((need.value,), device=x.device, dtype=torch.uint8)
            slot32 = _slot_table(slots, x.device, torch.int32)
            call("nrhip_hashgrid_multi_bwd_binned", C.byref(g), n_grids, _ptr(grid_id), _ptr(slot32), n_slots, _ptr(x),
                 _ptr(grad_out), n, _ptr(block), 1 if half else 0, _ptr(ws), ws.numel(), _stream())
            return block if block.dtype == out_dtype else block.to(out_dtype)
    # one zero-filled block for all touched grids (a scene has ~100 actor grids: one fill, not one per grid)
    flat = torch.zeros((n_slots, spec.table_rows, spec.features_per_level), device=x.device, dtype=torch.float32)
    g = spec.c_grid(flat[0])
    # device array of the gradient tables' addresses, computed ON the device (base + slot * stride; 0 for untouched grids):
    # a torch.tensor(list, device=...) here is a pageable host->device copy, i.e. a stream synchronisation per backward
    slot = _slot_table(slots, x.device, torch.int64)
    ptrs = torch.where(slot >= 0, slot * (flat[0].numel() * 4) + flat.data_ptr(), torch.zeros_like(slot))
    call("nrhip_hashgrid_multi_bwd", C.byref(g), n_grids, _ptr(grid_id), _ptr(x), _ptr(grad_out), x.shape[0], _ptr(ptrs),
         _stream())
    # fp16-storage grids: autograd wants the parameter's dtype -- ONE cast of the block
    return flat if out_dtype == torch.float32 else flat.to(out_dtype)


_SLOT_TABLES: dict = {}


def _slot_table(slots: Tuple[int, ...], device, dtype) -> Tensor:
    """int64 / int32 [n_grids] on the device: position of each grid in the gradient block, -1 for grids without one; uploaded
    once per distinct pattern"""
    key = (device, slots, dtype)
    t = _SLOT_TABLES.get(key)
    if t is None:
        if len(_SLOT_TABLES) >= 64:
            _SLOT_TABLES.clear()
        t = _SLOT_TABLES[key] = torch.tensor(slots, dtype=dtype, device=device)
    return t


def hashgrid_multi_bwd_input(spec: GridSpec, tables: Sequence[Tensor], grid_id: Tensor, x: Tensor, grad_out: Tensor):
    x, grid_id, grad_out = _chk(x, "x"), _chk(grid_id, "grid_id", torch.int32), _chk(grad_out, "grad_out")
    tables = [_chk(t, "table", tables[0].dtype) for t in tables]
    gx = torch.empty_like(x)
    g = spec.c_grid(tables[0])
    ptrs = _ptr_array(tables)
    call("nrhip_hashgrid_multi_bwd_input", C.byref(g), _ptr(ptrs), len(tables), _ptr(grid_id), _ptr(x), _ptr(grad_out),
         x.shape[0], _ptr(gx), _stream())
    return gx


def hashgrid_bwd_input(spec: GridSpec, table: Tensor, x: Tensor, grad_out: Tensor) -> Tensor:
    x, grad_out = _chk(x, "x"), _chk(grad_out, "grad_out")
    gx = torch.empty_like(x)
    g = spec.c_grid(table)
    call("nrhip_hashgrid_bwd_input", C.byref(g), _ptr(table), _ptr(x), _ptr(grad_out), x.shape[0], _ptr(gx), _stream())
    return gx


# A/B switch for profiling and for the parity test of the atomic path
_FORCE_ATOMIC_SCATTER = os.environ.get("NRHIP_ENCODE_BWD_ATOMIC") is not None
_BINNED_MIN_SAMPLES = 1 << 15
_MULTI_BWD_BINNED = os.environ.get("NRHIP_MULTI_BWD_BINNED", "1") != "0"  # 0: the actor grids' gradients by atomics (A/B)


def _table_grad_workspace(c_grid, n_samples: int, device) -> Optional[Tensor]:
    """Scratch for the atomics-free table gradients (csrc/encode_bwd_binned.hip); None -> use the atomic entry point."""
    if _FORCE_ATOMIC_SCATTER or n_samples < _BINNED_MIN_SAMPLES:
        return None  # small batches (one actor's hits): four launches + scratch cost more than the few atomics
    need = C.c_int64(0)
    call("nrhip_encode_bwd_binned_workspace", C.byref(c_grid), int(n_samples), C.byref(need))
    if need.value <= 0:
        return None
    return torch.empty((need.value,), device=device, dtype=torch.uint8)


def encode_fwd(spec: GridSpec, table: Tensor, static_scale: float, origins, directions, pixel_area, starts, ends):
    r, keep = _c_rays(origins, directions, pixel_area, starts, ends)
    out = torch.empty((r.n_rays * r.n_samples, spec.out_dim), device=origins.device, dtype=torch.float32)
    g = spec.c_grid(table)
    call("nrhip_encode_fwd", C.byref(g), _ptr(table), float(static_scale), C.byref(r), _ptr(out), _stream())
    return out


_BINNED_ROUND_SAMPLES = 1 << 23  # round_samples() of csrc/encode_bwd_binned.hip


def encode_bwd(spec: GridSpec, static_scale: float, origins, directions, pixel_area, starts, ends, grad_out,
               out_dtype=torch.float32):
    """-> grad table [L*T, F].  out_dtype=torch.float16 (an fp16-storage table): the binned path writes the gradient in fp16
    itself where it can (one round, i.e. <= 2^23 samples) -- otherwise fp32 is returned and the caller casts."""
    r, keep = _c_rays(origins, directions, pixel_area, starts, ends)
    grad_out = _chk(grad_out, "grad_out")
    n = r.n_rays * r.n_samples
    half = (out_dtype == torch.float16 and n <= _BINNED_ROUND_SAMPLES and not _FORCE_ATOMIC_SCATTER and n >= _BINNED_MIN_SAMPLES
            and "NRHIP_BIN_ROUND_LOG2" not in os.environ)
    gt = torch.empty((spec.table_rows, spec.features_per_level), device=origins.device,
                     dtype=torch.float16 if half else torch.float32)
    g = spec.c_grid(gt)
    ws = _table_grad_workspace(g, n, origins.device)
    if ws is not None and half:  # the fp16 gradient of an fp16-storage table, written by the partition itself
        call("nrhip_encode_bwd_binned_f16", C.byref(g), float(static_scale), C.byref(r), _ptr(grad_out), _ptr(gt), _ptr(ws),
             ws.numel(), _stream())
    elif ws is not None:  # overwrite = 1: the partition writes every element of the gradient, no zero-fill
        call("nrhip_encode_bwd_binned", C.byref(g), float(static_scale), C.byref(r), _ptr(grad_out), _ptr(gt), 1,
             _ptr(ws), ws.numel(), _stream())
    else:  # tables too large to cut into LDS slices, or a tiny batch: memory-side atomics (fp32 only)
        if gt.dtype != torch.float32:
            gt = torch.empty((spec.table_rows, spec.features_per_level), device=origins.device, dtype=torch.float32)
            g = spec.c_grid(gt)
        call("nrhip_encode_bwd", C.byref(g), float(static_scale), C.byref(r), _ptr(grad_out), _ptr(gt.zero_()),
             _stream())
    return gt


def encode_bwd_rays(spec: GridSpec, table: Tensor, static_scale: float, origins, directions, pixel_area, starts, ends,
                    grad_out) -> Tuple[Tensor, Tensor]:
    """dL/d(origins), dL/d(directions) [R,3] of the static encoding path from dL/d(rescaled features) [N, L*F]: the
    gradient a camera optimizer that moves the rays receives (cameras/camera_optimizers.py:173-182).  One kernel, no
    atomics (csrc/hashgrid_dx.hip)."""
    r, keep = _c_rays(origins, directions, pixel_area, starts, ends)
    grad_out = _chk(grad_out, "grad_out")
    if grad_out.numel() != r.n_rays * r.n_samples * spec.out_dim:
        raise ValueError(f"grad_out has {grad_out.numel()} elements, expected {r.n_rays * r.n_samples * spec.out_dim}")
    g = spec.c_grid(table)
    out = torch.empty((2, r.n_rays, 3), device=origins.device, dtype=torch.float32)
    call("nrhip_encode_bwd_rays", C.byref(g), _ptr(_chk(table, "table", table.dtype)), float(static_scale), C.byref(r),
         _ptr(grad_out), _ptr(out[0]), _ptr(out[1]), _stream())
    return out[0], out[1]


def sh4_fwd(dirs01: Tensor) -> Tensor:
    d = _chk(dirs01, "dirs")
    out = torch.empty((d.shape[0], 16), device=d.device, dtype=torch.float32)
    call("nrhip_sh4_fwd", _ptr(d), d.shape[0], _ptr(out), _stream())
    return out


def mlp_fwd(x: Tensor, weights, biases, save_hidden: bool = False):
    x = _chk(x, "x")
    m, keep = _c_mlp(weights, biases)
    if x.dim() != 2 or x.shape[1] != m.in_dim:
        raise ValueError(f"x must be [N,{m.in_dim}], got {tuple(x.shape)}")
    n = x.shape[0]
    y = torch.empty((n, m.out_dim), device=x.device, dtype=torch.float32)
    hidden = None
    if save_hidden and m.num_layers > 1:
        hidden = torch.empty((n, (m.num_layers - 1) * m.hidden_dim), device=x.device, dtype=torch.float32)
    call("nrhip_mlp_fwd", C.byref(m), _ptr(x), n, _ptr(y), _ptr(hidden), _stream())
    return (y, hidden) if save_hidden else y


def mlp_bwd(x: Tensor, hidden: Optional[Tensor], grad_y: Tensor, weights, biases, need_grad_x: bool = True):
    x, grad_y = _chk(x, "x"), _chk(grad_y, "grad_y")
    m, keep = _c_mlp(weights, biases)
    n = x.shape[0]
    gx = torch.empty_like(x) if need_grad_x else None
    # one zero-filled buffer for every weight / bias gradient of the MLP (one fill launch instead of 2 per layer)
    sizes = [w.numel() for w in weights] + [0 if b is None else b.numel() for b in biases]
    flat = torch.zeros((sum(sizes),), device=x.device, dtype=torch.float32)
    views = torch.split(flat, sizes)
    gws = [v.view_as(w) for v, w in zip(views[:len(weights)], weights)]
    gbs = [None if b is None else v.view_as(b) for v, b in zip(views[len(weights):], biases)]
    need = C.c_int64(0)
    call("nrhip_mlp_bwd_workspace", C.byref(m), n, C.byref(need))
    ws = torch.empty((max(need.value, 1),), device=x.device, dtype=torch.float32)
    pw = (C.c_void_p * _lib.MAX_LAYERS)(*[g.data_ptr() for g in gws])
    pb = (C.c_void_p * _lib.MAX_LAYERS)(*[(0 if g is None else g.data_ptr()) for g in gbs])
    call("nrhip_mlp_bwd", C.byref(m), _ptr(x), _ptr(hidden), _ptr(grad_y), n, _ptr(gx),
         C.cast(pw, C.POINTER(C.c_void_p)), C.cast(pb, C.POINTER(C.c_void_p)), _ptr(ws), need.value, _stream())
    return gx, gws, gbs


def field_feature_bwd_supported(weights, biases) -> bool:
    """shapes nrhip_field_feature_bwd covers: NeuRADField's feature head 48 -> {32,64} -> {32,64} -> 32 with biases"""
    return (len(weights) == 3 and weights[0].shape[1] == 48 and weights[2].shape[0] == 32 and weights[0].shape[0] in (32, 64)
            and all(b is not None for b in biases))


def field_feature_bwd(x: Tensor, hidden: Tensor, grad_feature: Tensor, grad_geo0: Tensor, weights, biases):
    """Backward of feature = geo[:, 1:] + mlp_feature([geo[:, 1:] | sh]) (neurad_field.py:146-152) in one pass: returns
    (grad_geo [N,33] = (grad_geo0 | grad_feature + grad_x[:, :32]), weight gradients, bias gradients)."""
    x, grad_feature, grad_geo0 = _chk(x, "x"), _chk(grad_feature, "grad_feature"), _chk(grad_geo0.reshape(-1), "grad_geo0")
    m, keep = _c_mlp(weights, biases)
    n = x.shape[0]
    if grad_geo0.shape[0] != n or grad_feature.shape != (n, 32):
        raise ValueError("field_feature_bwd: grad_feature [N,32] and grad_geo0 [N] expected")
    g_geo = torch.empty((n, 33), device=x.device, dtype=torch.float32)
    sizes = [w.numel() for w in weights] + [b.numel() for b in biases]
    flat = torch.zeros((sum(sizes),), device=x.device, dtype=torch.float32)
    views = torch.split(flat, sizes)
    gws = [v.view_as(w) for v, w in zip(views[:len(weights)], weights)]
    gbs = [v.view_as(b) for v, b in zip(views[len(weights):], biases)]
    need = C.c_int64(0)
    call("nrhip_mlp_bwd_workspace", C.byref(m), n, C.byref(need))
    ws = torch.empty((max(need.value, 1),), device=x.device, dtype=torch.float32)
    pw = (C.c_void_p * _lib.MAX_LAYERS)(*[g.data_ptr() for g in gws])
    pb = (C.c_void_p * _lib.MAX_LAYERS)(*[g.data_ptr() for g in gbs])
    call("nrhip_field_feature_bwd", C.byref(m), _ptr(x), _ptr(hidden), _ptr(grad_feature), _ptr(grad_geo0), n, _ptr(g_geo),
         C.cast(pw, C.POINTER(C.c_void_p)), C.cast(pb, C.POINTER(C.c_void_p)), _ptr(ws), need.value, _stream())
    return g_geo, gws, gbs


# ------------------------------------------------------------------------------------------------
@dataclass
class FieldSpec:
    """What the fused field/render kernels need (NeuRADField, neurad_field.py:78-152)."""

    grid: GridSpec
    table: Tensor
    static_scale: float
    geo_w: List[Tensor]
    geo_b: List[Optional[Tensor]]
    feat_w: List[Tensor]
    feat_b: List[Optional[Tensor]]
    use_sdf: bool = True
    beta: float = 20.0 + 1e-4  # |beta| + beta_min (model_components/utils.py:38-41)

    def c_field(self, eval_layout: bool = False):
        f = _lib.Field()
        f.grid = self.grid.c_grid(self.table)
        f.table = self.table.data_ptr()
        f.static_scale = float(self.static_scale)
        f.geo, k1 = _c_mlp(self.geo_w, self.geo_b)
        f.feat, k2 = _c_mlp(self.feat_w, self.feat_b)
        f.use_sdf = 1 if self.use_sdf else 0
        f.beta = float(self.beta)
        k3 = None
        if eval_layout:  # inference: the coarse levels from their shadow copies (bit-identical outputs)
            k3 = eval_table(self.grid, self.table)
            if k3 is not None:
                f.eval_table, f.eval_layout = k3[0].data_ptr(), k3[1]
        return f, (k1, k2, k3)


# Eval-time layout of the coarse levels (csrc/eval_layout.hip).  OPT-IN (NRHIP_EVAL_RELAYOUT=1): measured on BASELINE
# config[1] it changes neither the fabric reads nor the L1 -> L2 requests of the render kernel (the coarse levels are L2
# resident either way and the 16 lanes that share a level walk consecutive samples of one ray, i.e. the same lines in both
# layouts) and costs 2.6 % of kernel time (168.7 -> 173.1 us: one more LDS read and two more live registers per level);
# outputs are bit-identical.  profiles/r03_eval_relayout.txt.
_EVAL_RELAYOUT = os.environ.get("NRHIP_EVAL_RELAYOUT", "0") == "1"
_EVAL_TABLES: dict = {}  # (data_ptr, version, dtype, shape, grid key) -> (eval table, layout array, n shadow levels)


def eval_layout_plan(spec: GridSpec, table_dtype=torch.float32):
    """host logic: -> (layout ctypes array [L*4] = {mulY, mulZ, mask, row0} per level, rows of the eval table, number of
    levels that get a shadow copy)"""
    g = _lib.Grid()
    g.num_levels, g.n_features, g.log2_table_size = spec.num_levels, spec.features_per_level, spec.log2_hashmap_size
    g.param_dtype = 1 if table_dtype == torch.float16 else 0
    for i, v in enumerate(spec.scalings.tolist()):
        g.scalings[i] = v
    lay = (C.c_uint32 * (4 * spec.num_levels))()
    rows = C.c_int64(0)
    call("nrhip_eval_layout_plan", C.byref(g), lay, C.byref(rows))
    n_shadow = sum(1 for l in range(spec.num_levels) if lay[4 * l] != 2654435761)
    return lay, rows.value, n_shadow


def eval_table(spec: GridSpec, table: Tensor):
    """The table re-laid out for inference, cached per (storage, in-place version): -> (eval table, layout) or None when no
    level qualifies or the layout is switched off.  A derived buffer: the parameter (and the state_dict) stay as they are;
    writes through ``table.data`` do not bump the version -- call ``clear_eval_tables()`` after such edits."""
    if not _EVAL_RELAYOUT or not table.is_cuda:
        return None
    key = (table.data_ptr(), table._version, table.dtype, tuple(table.shape), spec.num_levels, spec.min_res, spec.max_res)
    hit = _EVAL_TABLES.get(key)
    if hit is None:
        lay, rows, n_shadow = eval_layout_plan(spec, table.dtype)
        if n_shadow == 0:
            hit = (None, None)
        else:
            out = torch.empty((rows, spec.features_per_level), device=table.device, dtype=table.dtype)
            g = spec.c_grid(table)
            call("nrhip_eval_layout_build", C.byref(g), _ptr(table), lay, _ptr(out), _stream())
            hit = (out, lay)
        for k in [k for k in _EVAL_TABLES if k[0] == key[0] and k != key]:  # older versions of the same parameter
            del _EVAL_TABLES[k]
        if len(_EVAL_TABLES) >= 4:
            _EVAL_TABLES.clear()
        _EVAL_TABLES[key] = hit
    return None if hit[0] is None else hit


TABLE_EPOCH = [0]  # bumped whenever a kernel has written tables behind autograd's back (the optimizer): copies made from
                   # the tables before that (eval re-layouts, field_components/neurad_encoding.py's stacked actor tables) are stale


def clear_eval_tables() -> None:
    _EVAL_TABLES.clear()
    TABLE_EPOCH[0] += 1


def field_fwd(fs: FieldSpec, origins, directions, pixel_area, starts, ends, order: Optional[Tensor] = None):
    """-> feature [R,S,32], sdf (or raw geo output) [R,S], alpha (or density) [R,S]"""
    r, keep = _c_rays(origins, directions, pixel_area, starts, ends, order)
    f, keep2 = fs.c_field()
    R, S = r.n_rays, r.n_samples
    dev = origins.device
    feature = torch.empty((R, S, 32), device=dev, dtype=torch.float32)
    sdf = torch.empty((R, S), device=dev, dtype=torch.float32)
    alpha = torch.empty((R, S), device=dev, dtype=torch.float32)
    call("nrhip_field_fwd", C.byref(f), C.byref(r), _ptr(feature), _ptr(sdf), _ptr(alpha), _stream())
    return feature, sdf, alpha


def field_fwd_train(fs: FieldSpec, origins, directions, pixel_area, starts, ends, order: Optional[Tensor] = None,
                    override=None):
    """field_fwd + the activations the backward needs: -> (feature [N,32], geo_out [N], head [N]),
    (enc [N,32], geo_hidden [N,H], feat_in [N,48], feat_hidden [N,2H]).
    override = (ovr_row int32 [N] (row index or -1), ovr_rows [P,32], ovr_dirs [P,3]): samples inside an actor box take
    their encoding row and SH direction from the caller (nrhip_field_fwd_train_ovr)."""
    r, keep = _c_rays(origins, directions, pixel_area, starts, ends, order)
    f, keep2 = fs.c_field()
    n, dev = r.n_rays * r.n_samples, origins.device
    H = fs.geo_w[0].shape[0]
    mk = lambda c: torch.empty((n, c), device=dev, dtype=torch.float32)  # noqa: E731
    feature, enc, hg, xf, hf = mk(32), mk(32), mk(H), mk(48), mk(2 * H)
    sdf = torch.empty((n,), device=dev, dtype=torch.float32)
    head = torch.empty((n,), device=dev, dtype=torch.float32)
    if override is not None:
        ov, rows, dirs = override
        ov, rows, dirs = _chk(ov.reshape(-1), "ovr_row", torch.int32), _chk(rows, "ovr_rows"), _chk(dirs, "ovr_dirs")
        if ov.shape[0] != n or rows.dim() != 2 or rows.shape[1] != 32 or dirs.shape != (rows.shape[0], 3):
            raise ValueError("field_fwd_train: override = (int32 [N], [P,32], [P,3])")
        call("nrhip_field_fwd_train_ovr", C.byref(f), C.byref(r), _ptr(ov), _ptr(rows), _ptr(dirs), _ptr(feature), _ptr(sdf),
             _ptr(head), _ptr(enc), _ptr(hg), _ptr(xf), _ptr(hf), _stream())
    else:
        call("nrhip_field_fwd_train", C.byref(f), C.byref(r), _ptr(feature), _ptr(sdf), _ptr(head), _ptr(enc), _ptr(hg),
             _ptr(xf), _ptr(hf), _stream())
    return (feature, sdf, head), (enc, hg, xf, hf)


def render_fwd(fs: FieldSpec, origins, directions, pixel_area, starts, ends, return_weights: bool = False,
               out: Optional[Tuple[Tensor, Tensor, Tensor]] = None, early_stop_eps: float = 0.0,
               order: Optional[Tensor] = None):
    """The fused headline kernel.  -> features [R,32], depth [R,1], accumulation [R,1] (, weights [R,S]).
    early_stop_eps > 0 (eval option, default exact): rays stop once their transmittance is below it.
    order: processing order from ``ray_order`` (locality hint; outputs stay in batch order)."""
    r, keep = _c_rays(origins, directions, pixel_area, starts, ends, order)
    f, keep2 = fs.c_field(eval_layout=not fs.table.requires_grad)
    R, S = r.n_rays, r.n_samples
    dev = origins.device
    if out is None:
        feats = torch.empty((R, 32), device=dev, dtype=torch.float32)
        depth = torch.empty((R, 1), device=dev, dtype=torch.float32)
        acc = torch.empty((R, 1), device=dev, dtype=torch.float32)
    else:
        feats, depth, acc = out
    w = torch.empty((R, S), device=dev, dtype=torch.float32) if return_weights else None
    call("nrhip_render_fwd_ex", C.byref(f), C.byref(r), _ptr(feats), _ptr(depth), _ptr(acc), _ptr(w),
         float(early_stop_eps), _stream())
    return (feats, depth, acc, w) if return_weights else (feats, depth, acc)


def render_fwd_actors(fs: FieldSpec, spec: "ActorSpec", cand, origins, directions, pixel_area, starts, ends,
                      return_weights: bool = False, early_stop_eps: float = 0.0, order: Optional[Tensor] = None):
    """``render_fwd`` for a scene with dynamic actors, still one kernel: samples inside an actor box read that actor's
    grid (cand = ``actor_prepare``'s per-ray candidate lists).  Raises NrhipError(UNSUPPORTED) when the actor grid does
    not share the static grid's features per level -- callers fall back to the operator-level path."""
    r, keep = _c_rays(origins, directions, pixel_area, starts, ends, order)
    f, keep2 = fs.c_field()
    a, keep3 = spec.c_actors()
    cnt, act, w2b, _ = cand
    R, S = r.n_rays, r.n_samples
    dev = origins.device
    feats = torch.empty((R, 32), device=dev, dtype=torch.float32)
    depth = torch.empty((R, 1), device=dev, dtype=torch.float32)
    acc = torch.empty((R, 1), device=dev, dtype=torch.float32)
    w = torch.empty((R, S), device=dev, dtype=torch.float32) if return_weights else None
    work = torch.empty((R + 4,), device=dev, dtype=torch.int32)
    call("nrhip_render_fwd_actors", C.byref(f), C.byref(a), C.byref(r), _ptr(cnt), _ptr(act), _ptr(w2b), _ptr(feats),
         _ptr(depth), _ptr(acc), _ptr(w), float(early_stop_eps), _ptr(work), _stream())
    return (feats, depth, acc, w) if return_weights else (feats, depth, acc)


# ------------------------------------------------------------------------------------------------
def render_weight_from_alpha(alphas: Tensor):
    a = _chk(alphas, "alphas")
    R, S = a.shape
    w, t = torch.empty_like(a), torch.empty_like(a)
    call("nrhip_render_weight_from_alpha", _ptr(a), R, S, _ptr(w), _ptr(t), _stream())
    return w, t


def render_weight_from_alpha_bwd(alphas, grad_w, grad_t=None):
    a, gw = _chk(alphas, "alphas"), _chk(grad_w, "grad_w")
    gt = None if grad_t is None else _chk(grad_t, "grad_t")
    ga = torch.empty_like(a)
    call("nrhip_render_weight_from_alpha_bwd", _ptr(a), _ptr(gw), _ptr(gt), a.shape[0], a.shape[1], _ptr(ga), _stream())
    return ga


def render_weight_from_density(t_starts, t_ends, sigmas):
    s, e, sg = _chk(t_starts, "t_starts"), _chk(t_ends, "t_ends"), _chk(sigmas, "sigmas")
    R, S = sg.shape
    w, t, a = torch.empty_like(sg), torch.empty_like(sg), torch.empty_like(sg)
    call("nrhip_render_weight_from_density", _ptr(s), _ptr(e), _ptr(sg), R, S, _ptr(w), _ptr(t), _ptr(a), _stream())
    return w, t, a


def render_weight_from_density_bwd(t_starts, t_ends, sigmas, grad_w):
    s, e, sg, gw = (_chk(v, n) for v, n in ((t_starts, "t_starts"), (t_ends, "t_ends"), (sigmas, "sigmas"),
                                            (grad_w, "grad_w")))
    gs = torch.empty_like(sg)
    call("nrhip_render_weight_from_density_bwd", _ptr(s), _ptr(e), _ptr(sg), _ptr(gw), sg.shape[0], sg.shape[1],
         _ptr(gs), _stream())
    return gs


def accumulate_along_rays(weights, values=None):
    w = _chk(weights, "weights")
    R, S = w.shape
    if values is None:
        out = torch.empty((R, 1), device=w.device, dtype=torch.float32)
        call("nrhip_accumulate_along_rays", _ptr(w), _ptr(None), R, S, 1, _ptr(out), _stream())
        return out
    v = _chk(values, "values")
    Cc = v.shape[-1]
    out = torch.empty((R, Cc), device=w.device, dtype=torch.float32)
    call("nrhip_accumulate_along_rays", _ptr(w), _ptr(v), R, S, Cc, _ptr(out), _stream())
    return out


def composite_fwd(weights, features, starts, ends):
    w, f, s, e = (_chk(v, n) for v, n in ((weights, "weights"), (features, "features"), (starts, "starts"),
                                          (ends, "ends")))
    R, S, Cc = f.shape
    of = torch.empty((R, Cc), device=w.device, dtype=torch.float32)
    od = torch.empty((R, 1), device=w.device, dtype=torch.float32)
    oa = torch.empty((R, 1), device=w.device, dtype=torch.float32)
    call("nrhip_composite_fwd", _ptr(w), _ptr(f), _ptr(s), _ptr(e), R, S, Cc, _ptr(of), _ptr(od), _ptr(oa), _stream())
    return of, od, oa


def lidar_carving(starts: Tensor, ends: Tensor, is_lidar: Tensor, did_return: Optional[Tensor], distance: Tensor,
                  carving_epsilon: float, non_return_lidar_distance: float, weights: Optional[Tensor] = None,
                  want_mask: bool = True, want_grad: bool = True):
    """starts/ends [R,S] (any row stride), per-ray is_lidar / did_return (bool) / distance -> (is_close [R,S] bool or
    None, loss_per_ray [R] or None, grad_weights [R,S] or None); the loss terms need ``weights`` [R,S]."""
    R, S = starts.shape
    assert starts.stride(1) == 1 and ends.stride(1) == 1 and starts.stride(0) == ends.stride(0)
    for v, n in ((starts, "starts"), (ends, "ends")):
        if not v.is_cuda or v.dtype != torch.float32:
            raise _lib.NeuradHipError(f"{n}: expected a float32 GPU tensor")
    u8 = lambda m: None if m is None else m.reshape(-1).contiguous().view(torch.uint8)  # noqa: E731  (bool is 1 byte)
    lid, ret = u8(is_lidar), u8(did_return)
    dist = _chk(distance.reshape(-1), "distance")
    w = None if weights is None else _chk(weights, "weights")
    dev = starts.device
    close = torch.empty((R, S), dtype=torch.bool, device=dev) if want_mask else None
    loss = torch.empty((R,), dtype=torch.float32, device=dev) if w is not None else None
    gw = torch.empty((R, S), dtype=torch.float32, device=dev) if (w is not None and want_grad) else None
    call("nrhip_lidar_carving", _ptr(starts), _ptr(ends), starts.stride(0), _ptr(w), _ptr(lid), _ptr(ret), _ptr(dist),
         float(carving_epsilon), float(non_return_lidar_distance), R, S, _ptr(close), _ptr(loss), _ptr(gw), _stream())
    return close, loss, gw


def embedding_lerp(weight: Tensor, idx_lo: Tensor, idx_hi: Optional[Tensor] = None, frac: Optional[Tensor] = None) -> Tensor:
    """out[r] = weight[idx_lo[r]] * (1 - frac[r]) + weight[idx_hi[r]] * frac[r]   (idx_hi None: weight[idx_lo])"""
    w, lo = _chk(weight, "weight"), _chk(idx_lo.reshape(-1), "idx_lo", torch.int64)
    hi = None if idx_hi is None else _chk(idx_hi.reshape(-1), "idx_hi", torch.int64)
    fr = None if frac is None else _chk(frac.reshape(-1), "frac")
    out = torch.empty((lo.shape[0], w.shape[1]), dtype=torch.float32, device=w.device)
    call("nrhip_embedding_lerp_fwd", _ptr(w), _ptr(lo), _ptr(hi), _ptr(fr), lo.shape[0], w.shape[0], w.shape[1], _ptr(out),
         _stream())
    return out


def embedding_lerp_bwd(g_out: Tensor, idx_lo, idx_hi, frac, n_embed: int) -> Tensor:
    g, lo = _chk(g_out, "g_out"), _chk(idx_lo.reshape(-1), "idx_lo", torch.int64)
    hi = None if idx_hi is None else _chk(idx_hi.reshape(-1), "idx_hi", torch.int64)
    fr = None if frac is None else _chk(frac.reshape(-1), "frac")
    gw = torch.zeros((n_embed, g.shape[1]), dtype=torch.float32, device=g.device)
    call("nrhip_embedding_lerp_bwd", _ptr(g), _ptr(lo), _ptr(hi), _ptr(fr), lo.shape[0], n_embed, g.shape[1], _ptr(gw),
         _stream())
    return gw


def accumulate_along_rays_bwd(weights, values, g_out, need_grad_weights=True, need_grad_values=True):
    """-> (grad weights [R,S] or None, grad values [R,S,C] or None) of out[r,c] = sum_s w[r,s] v[r,s,c]"""
    w, v, g = _chk(weights, "weights"), _chk(values, "values"), _chk(g_out, "g_out")
    R, S, Cc = v.shape
    gw = torch.empty_like(w) if need_grad_weights else None
    gv = torch.empty_like(v) if need_grad_values else None
    call("nrhip_accumulate_along_rays_bwd", _ptr(w), _ptr(v), _ptr(g), R, S, Cc, _ptr(gw), _ptr(gv), _stream())
    return gw, gv


def composite_bwd(weights, features, starts, ends, g_feat, g_depth=None, g_acc=None, need_grad_features=True):
    w, f, s, e, gf = (_chk(v, n) for v, n in ((weights, "weights"), (features, "features"), (starts, "starts"),
                                              (ends, "ends"), (g_feat, "g_feat")))
    gd = None if g_depth is None else _chk(g_depth.reshape(-1), "g_depth")
    ga = None if g_acc is None else _chk(g_acc.reshape(-1), "g_acc")
    R, S, Cc = f.shape
    gw = torch.empty_like(w)
    gfe = torch.empty_like(f) if need_grad_features else None
    call("nrhip_composite_bwd", _ptr(w), _ptr(f), _ptr(s), _ptr(e), _ptr(gf), _ptr(gd), _ptr(ga), R, S, Cc, _ptr(gw),
         _ptr(gfe), _stream())
    return gw, gfe


# ------------------------------------------------------------------------------------------------
@dataclass
class ProposalSpec:
    grid: GridSpec
    table: Tensor
    static_scale: float
    decoder_weight: Tensor  # [1, L] (nn.Linear(L,1,bias=False).weight)

    def c_prop(self):
        p = _lib.Proposal()
        p.grid = self.grid.c_grid(self.table)
        p.table = self.table.data_ptr()
        p.static_scale = float(self.static_scale)
        dw = _chk(self.decoder_weight.reshape(-1), "decoder_weight")
        p.decoder_weight = dw.data_ptr()
        return p, dw


def proposal_density_fwd(ps: ProposalSpec, origins, directions, pixel_area, starts, ends, save_features: bool = False):
    """density [R,S]; with save_features also the rescaled per-level features, level-major [L, R*S], for the backward"""
    r, keep = _c_rays(origins, directions, pixel_area, starts, ends)
    p, keep2 = ps.c_prop()
    dens = torch.empty((r.n_rays, r.n_samples), device=origins.device, dtype=torch.float32)
    lf = (torch.empty((ps.grid.num_levels, r.n_rays * r.n_samples), device=origins.device, dtype=torch.float32)
          if save_features else None)
    call("nrhip_proposal_density_fwd", C.byref(p), C.byref(r), _ptr(dens), _ptr(lf), _stream())
    return (dens, lf) if save_features else dens


def proposal_density_bwd(ps: ProposalSpec, origins, directions, pixel_area, starts, ends, density, grad_density,
                         level_features: Optional[Tensor] = None):
    r, keep = _c_rays(origins, directions, pixel_area, starts, ends)
    p, keep2 = ps.c_prop()
    gdec = torch.zeros((1, ps.grid.num_levels), device=origins.device, dtype=torch.float32)
    ws = _table_grad_workspace(p.grid, r.n_rays * r.n_samples, origins.device)
    if ws is not None:
        gt = torch.empty((ps.grid.table_rows, 1), device=origins.device, dtype=torch.float32)
        call("nrhip_proposal_density_bwd_binned", C.byref(p), C.byref(r), _ptr(_chk(density, "density")),
             _ptr(level_features), _ptr(_chk(grad_density, "grad_density")), _ptr(gt), _ptr(gdec), 1, _ptr(ws),
             ws.numel(), _stream())
    else:
        if ps.table.dtype != torch.float32:  # the atomic kernel recomputes the features from an fp32 table: small batches only
            ps32 = ProposalSpec(ps.grid, ps.table.float(), ps.static_scale, ps.decoder_weight)  # (alive until the launch)
            p, keep2 = ps32.c_prop()
        gt = torch.zeros((ps.grid.table_rows, 1), device=origins.device, dtype=torch.float32)
        call("nrhip_proposal_density_bwd", C.byref(p), C.byref(r), _ptr(_chk(density, "density")),
             _ptr(_chk(grad_density, "grad_density")), _ptr(gt), _ptr(gdec), _stream())
    return gt, gdec


def weights_from_density(deltas, densities) -> Tensor:
    d, s = _chk(deltas, "deltas"), _chk(densities, "densities")
    w = torch.empty_like(s)
    call("nrhip_weights_from_density", _ptr(d), _ptr(s), s.shape[0], s.shape[1], _ptr(w), _stream())
    return w


def weights_from_density_bwd(deltas, densities, grad_w) -> Tensor:
    d, s, g = _chk(deltas, "deltas"), _chk(densities, "densities"), _chk(grad_w, "grad_w")
    gs = torch.empty_like(s)
    call("nrhip_weights_from_density_bwd", _ptr(d), _ptr(s), _ptr(g), s.shape[0], s.shape[1], _ptr(gs), _stream())
    return gs


def power_sampler(nears: Optional[Tensor], fars: Tensor, num_samples: int, lam: float = -1.0, scaling: float = 0.1,
                  t_rand: Optional[Tensor] = None, last_edge: float = 0.0):
    """-> spacing bins, euclidean bins [R,S+1]; last_edge > 0 sets the last euclidean edge (the model's sky stretch)"""
    f = _chk(fars.reshape(-1), "fars")
    n = None if nears is None else _chk(nears.reshape(-1), "nears")
    R = f.shape[0]
    tr = None if t_rand is None else _chk(t_rand, "t_rand")
    sp = torch.empty((R, num_samples + 1), device=f.device, dtype=torch.float32)
    eu = torch.empty_like(sp)
    call("nrhip_power_sampler", _ptr(n), _ptr(f), R, num_samples, float(lam), float(scaling), _ptr(tr),
         float(last_edge), _ptr(sp), _ptr(eu), _stream())
    return sp, eu


def power_sampler_ordered(nears: Optional[Tensor], fars: Tensor, num_samples: int, origins: Tensor, directions: Tensor,
                          static_scale: float, lam: float = -1.0, scaling: float = 0.1, t_rand: Optional[Tensor] = None,
                          last_edge: float = 0.0, t_ref: Optional[float] = None, key_bits: int = 0):
    """``power_sampler`` and ``ray_order`` as one launch -> (spacing bins, euclidean bins [R,S+1], order int32 [R])"""
    f = _chk(fars.reshape(-1), "fars")
    n = None if nears is None else _chk(nears.reshape(-1), "nears")
    o, d = _chk(origins, "origins"), _chk(directions, "directions")
    R = f.shape[0]
    if o.shape != (R, 3) or d.shape != (R, 3):
        raise ValueError(f"origins / directions must be [R={R},3]")
    tr = None if t_rand is None else _chk(t_rand, "t_rand")
    sp = torch.empty((R, num_samples + 1), device=f.device, dtype=torch.float32)
    eu = torch.empty_like(sp)
    order = torch.empty((R,), device=f.device, dtype=torch.int32)
    call("nrhip_power_sampler_ordered", _ptr(n), _ptr(f), R, num_samples, float(lam), float(scaling), _ptr(tr),
         float(last_edge), _ptr(sp), _ptr(eu), _ptr(o), _ptr(d), float(static_scale if t_ref is None else t_ref),
         float(static_scale), int(key_bits), _ptr(order), _stream())
    return sp, eu, order


def pdf_sample(weights, spacing_bins, nears, fars, num_samples, lam=-1.0, scaling=0.1, histogram_padding=0.01,
               rand: Optional[Tensor] = None):
    w, b = _chk(weights, "weights"), _chk(spacing_bins, "spacing_bins")
    f = _chk(fars.reshape(-1), "fars")
    n = None if nears is None else _chk(nears.reshape(-1), "nears")
    R, Sp = w.shape
    stride = 0
    if rand is not None:
        rand = _chk(rand, "rand")
        stride = 0 if rand.numel() == R else num_samples + 1
    sp = torch.empty((R, num_samples + 1), device=w.device, dtype=torch.float32)
    eu = torch.empty_like(sp)
    call("nrhip_pdf_sample", _ptr(w), _ptr(b), _ptr(n), _ptr(f), R, Sp, num_samples, float(lam), float(scaling),
         float(histogram_padding), _ptr(rand), stride, _ptr(sp), _ptr(eu), _stream())
    return sp, eu


def proposal_sampler_fwd(props: Sequence[ProposalSpec], origins, directions, pixel_area, nears, fars,
                         num_samples=(128, 64, 32), lam=-1.0, scaling=0.1, histogram_padding=0.01,
                         sky_distance=20000.0, actor_specs: Optional[Sequence["ActorSpec"]] = None, cand=None):
    """Fused S5 (+ the far clamp of M1).  -> (weights per round, spacing bins per round+1, euclid bins per round+1).
    actor_specs[i] (+ cand, the per-ray candidate lists of ``actor_prepare``): the actor grids of props[i] -- proposal
    samples inside an actor box take their density from them (nrhip_proposal_sampler_fwd_actors)."""
    n_rounds = len(props)
    if len(num_samples) != n_rounds + 1:
        raise ValueError("num_samples needs one entry per proposal round plus the final count")
    o, d = _chk(origins, "origins"), _chk(directions, "directions")
    a = _chk(pixel_area.reshape(-1), "pixel_area")
    f = None if fars is None else _chk(fars.reshape(-1), "fars")
    n = None if nears is None else _chk(nears.reshape(-1), "nears")
    R = o.shape[0]
    cfg = _lib.SamplerCfg()
    cfg.n_rounds = n_rounds
    for i, v in enumerate(num_samples):
        cfg.n_samples[i] = v
    cfg.lam, cfg.scaling, cfg.histogram_padding, cfg.sky_distance = lam, scaling, histogram_padding, sky_distance
    cprops = (_lib.Proposal * n_rounds)()
    keep = []
    for i, p in enumerate(props):
        cp, k = p.c_prop()
        cprops[i] = cp
        keep.append(k)
    ws = [torch.empty((R, num_samples[i]), device=o.device, dtype=torch.float32) for i in range(n_rounds)]
    sps = [torch.empty((R, num_samples[i] + 1), device=o.device, dtype=torch.float32) for i in range(n_rounds + 1)]
    eus = [torch.empty((R, num_samples[i] + 1), device=o.device, dtype=torch.float32) for i in range(n_rounds + 1)]
    pw = (C.c_void_p * n_rounds)(*[t.data_ptr() for t in ws])
    psp = (C.c_void_p * (n_rounds + 1))(*[t.data_ptr() for t in sps])
    peu = (C.c_void_p * (n_rounds + 1))(*[t.data_ptr() for t in eus])
    if actor_specs is not None:
        cacts = (_lib.Actors * n_rounds)()
        for i, s in enumerate(actor_specs):
            ca, k = s.c_actors()
            cacts[i] = ca
            keep.append(k)
        cnt, act, w2b, _ = cand
        call("nrhip_proposal_sampler_fwd_actors", C.byref(cfg), cprops, cacts, _ptr(cnt), _ptr(act), _ptr(w2b), _ptr(o),
             _ptr(d), _ptr(a), _ptr(n), _ptr(f), R, C.cast(pw, C.POINTER(C.c_void_p)), C.cast(psp, C.POINTER(C.c_void_p)),
             C.cast(peu, C.POINTER(C.c_void_p)), _stream())
        return ws, sps, eus
    call("nrhip_proposal_sampler_fwd", C.byref(cfg), cprops, _ptr(o), _ptr(d), _ptr(a), _ptr(n), _ptr(f), R,
         C.cast(pw, C.POINTER(C.c_void_p)), C.cast(psp, C.POINTER(C.c_void_p)), C.cast(peu, C.POINTER(C.c_void_p)),
         _stream())
    return ws, sps, eus


@dataclass
class ActorSpec:
    """Device-side view of DynamicActors + the per-actor grids."""

    timestamps: Tensor      # [Tn] fp32
    positions: Tensor       # [Tn,A,3]
    rotations_6d: Tensor    # [Tn,A,6]
    present: Tensor         # [Tn,A] bool
    bounds: Tensor          # [A,3]
    grid: GridSpec
    tables: List[Tensor]    # A tables [L*T, F] fp32, already ordered by actor_to_id
    actor_scale: float = 10.0

    def c_actors(self):
        # plain pointers + sizes: valid while this spec's tensors are the same storage and `present` is unedited -- re-built
        # when one of them is re-assigned or `present` is written in place
        key = (self.timestamps.data_ptr(), self.positions.data_ptr(), self.rotations_6d.data_ptr(), self.present.data_ptr(),
               self.present._version, self.bounds.data_ptr(), tuple(t.data_ptr() for t in self.tables), self.actor_scale)
        cached = getattr(self, "_c_actors", None)
        if cached is not None and cached[0] == key:
            return cached[1]
        present = self.present.contiguous()
        present = present.view(torch.uint8) if present.dtype == torch.bool else present.to(torch.uint8)  # bool is 1 byte
        keep = [_chk(self.timestamps, "timestamps"), _chk(self.positions, "positions"),
                _chk(self.rotations_6d, "rotations_6d"), present, _chk(self.bounds, "bounds")]
        tabs = [_chk(t, "actor table", self.tables[0].dtype) for t in self.tables]  # one storage type (fp32 | fp16)
        # device array of table pointers: uploaded once per set of tables, not once per call (a pageable H2D copy
        # synchronises the host with the stream)
        ptrs = _ptr_array(tabs)
        a = _lib.Actors()
        a.n_times, a.n_actors = self.positions.shape[0], self.positions.shape[1]
        a.timestamps, a.positions, a.rotations_6d = keep[0].data_ptr(), keep[1].data_ptr(), keep[2].data_ptr()
        a.present, a.bounds = keep[3].data_ptr(), keep[4].data_ptr()
        a.grid = self.grid.c_grid(tabs[0])
        a.tables = ptrs.data_ptr()
        a.actor_scale = float(self.actor_scale)
        a.max_candidates = a.n_actors  # per-ray lists as long as the actor count: no ray can overflow, no host check
        self._c_actors = (key, (a, (keep, tabs, ptrs)))
        return self._c_actors[1]


def actor_prepare(spec: ActorSpec, origins, directions, pixel_area, starts, ends, times, edit: Optional[dict] = None):
    """-> (cand_count [R] i32, cand_actor [R,K] i32, cand_w2b [R,K,12], None) with K = the number of actors, so every
    actor a ray passes fits (the reference has no limit either); R*K*52 bytes, e.g. 300 MB for 57 344 rays x 100 actors.
    edit: DynamicActors.actor_editing (lateral / longitudinal / height / rotation / index, dynamic_actors.py:53-59) for an
    eval-time move of the boxes (nrhip_actor_prepare_edited), None: the trajectories as they are."""
    r, keep = _c_rays(origins, directions, pixel_area, starts, ends)
    a, keep2 = spec.c_actors()
    R, K, dev = r.n_rays, a.max_candidates, origins.device
    t = _chk(times.reshape(-1), "times")
    cnt = torch.empty((R,), dtype=torch.int32, device=dev)
    act = torch.empty((R, K), dtype=torch.int32, device=dev)  # only the first cnt[r] entries of a row are ever read
    w2b = torch.empty((R, K, 12), dtype=torch.float32, device=dev)
    if edit is None:
        call("nrhip_actor_prepare", C.byref(a), C.byref(r), _ptr(t), _ptr(cnt), _ptr(act), _ptr(w2b), _ptr(None), _stream())
    else:
        e = _lib.ActorEdit(float(edit.get("lateral", 0.0)), float(edit.get("longitudinal", 0.0)),
                           float(edit.get("height", 0.0)), float(edit.get("rotation", 0.0)), int(edit.get("index", -1)))
        call("nrhip_actor_prepare_edited", C.byref(a), C.byref(r), _ptr(t), C.byref(e), _ptr(cnt), _ptr(act), _ptr(w2b),
             _ptr(None), _stream())
    return cnt, act, w2b, None


def actor_encode(spec: ActorSpec, cand, origins, directions, pixel_area, starts, ends, features: Tensor,
                 ray_flip: Optional[Tensor] = None):
    """Overwrites the rows of ``features`` [N,out_dim] whose sample lies inside an actor box (in place).
    -> (directions [N,3], hit_actor [N] int32: actor index or -1)"""
    r, keep = _c_rays(origins, directions, pixel_area, starts, ends)
    a, keep2 = spec.c_actors()
    cnt, act, w2b, _ = cand
    n = r.n_rays * r.n_samples
    feats = _chk(features, "features")
    assert feats.data_ptr() == features.data_ptr(), "features must be contiguous (updated in place)"
    dirs = torch.empty((n, 3), dtype=torch.float32, device=feats.device)
    hit = torch.empty((n,), dtype=torch.int32, device=feats.device)
    call("nrhip_actor_encode", C.byref(a), C.byref(r), _ptr(cnt), _ptr(act), _ptr(w2b), feats.shape[1], _ptr(feats),
         _ptr(dirs), _ptr(hit), _ptr(None if ray_flip is None else _chk(ray_flip.reshape(-1), "ray_flip")), _stream())
    return dirs, hit  # int32: actor index or -1


def actor_pair_positions(spec: ActorSpec, origins, directions, pixel_area, starts, ends, times, sample_idx: Tensor,
                         actor_idx: Tensor, ray_flip: Optional[Tensor] = None):
    """Box-frame, contracted position of (sample, actor) pairs.  sample_idx [P] int64 flat sample index, actor_idx [P]
    int32.  -> (x01 [P,3] in [0,1]^3, cstd [P])"""
    r, keep = _c_rays(origins, directions, pixel_area, starts, ends)
    a, keep2 = spec.c_actors()
    si, ai = _chk(sample_idx, "sample_idx", torch.int64), _chk(actor_idx, "actor_idx", torch.int32)
    P_ = si.shape[0]
    x01 = torch.empty((P_, 3), dtype=torch.float32, device=si.device)
    cstd = torch.empty((P_,), dtype=torch.float32, device=si.device)
    call("nrhip_actor_pair_positions_fwd", C.byref(a), C.byref(r), _ptr(_chk(times.reshape(-1), "times")), _ptr(si),
         _ptr(ai), _ptr(None if ray_flip is None else _chk(ray_flip.reshape(-1), "ray_flip")), P_, _ptr(x01), _ptr(cstd),
         _stream())
    return x01, cstd


def actor_pair_positions_bwd(spec: ActorSpec, origins, directions, pixel_area, starts, ends, times, sample_idx, actor_idx,
                             ray_flip, grad_x01: Tensor, grad_cstd: Tensor, ray_grads: bool = False):
    """-> (grad actor_positions [Tn,A,3], grad actor_rotations_6d [Tn,A,6]); with ``ray_grads`` also (grad origins [R,3],
    grad directions [R,3]): the in-box samples' world positions move with the ray (camera optimizer)"""
    r, keep = _c_rays(origins, directions, pixel_area, starts, ends)
    a, keep2 = spec.c_actors()
    si, ai = _chk(sample_idx, "sample_idx", torch.int64), _chk(actor_idx, "actor_idx", torch.int32)
    flat = torch.zeros((a.n_times * a.n_actors * 9,), dtype=torch.float32, device=si.device)
    gp = flat[: a.n_times * a.n_actors * 3].view(a.n_times, a.n_actors, 3)
    gr = flat[a.n_times * a.n_actors * 3:].view(a.n_times, a.n_actors, 6)
    common = (C.byref(a), C.byref(r), _ptr(_chk(times.reshape(-1), "times")), _ptr(si), _ptr(ai),
              _ptr(None if ray_flip is None else _chk(ray_flip.reshape(-1), "ray_flip")), si.shape[0],
              _ptr(_chk(grad_x01, "grad_x01")), _ptr(_chk(grad_cstd, "grad_cstd")), _ptr(gp), _ptr(gr))
    if not ray_grads:
        call("nrhip_actor_pair_positions_bwd", *common, _stream())
        return gp, gr
    god = torch.zeros((2, r.n_rays, 3), dtype=torch.float32, device=si.device)
    call("nrhip_actor_pair_positions_bwd_rays", *common, _ptr(god[0]), _ptr(god[1]), _stream())
    return gp, gr, god[0], god[1]


def actor_hits(spec: ActorSpec, cand, origins, directions, pixel_area, starts, ends) -> Tensor:
    """-> hits [N,8] int32: the actors whose boxes contain the sample, ascending, padded with -1"""
    r, keep = _c_rays(origins, directions, pixel_area, starts, ends)
    a, keep2 = spec.c_actors()
    cnt, act, w2b, _ = cand
    hits = torch.empty((r.n_rays * r.n_samples, _lib.MAX_SAMPLE_CONTAINMENTS), dtype=torch.int32, device=origins.device)
    call("nrhip_actor_hits", C.byref(a), C.byref(r), _ptr(cnt), _ptr(act), _ptr(w2b), _ptr(hits), _stream())
    return hits


def actor_pairs(hits: Tensor) -> Tuple[Tensor, Tensor]:
    """-> (sample_idx int64 [P], actor_idx int32 [P]): every (sample, containing actor) of a hits table in (sample, slot)
    order -- `(hits >= 0).nonzero()` + `hits[idx, slot]` without the rocprim partition, the gather and the casts (one host
    read of P, which nonzero needs as well)"""
    h = _chk(hits, "hits", torch.int32)
    n = h.shape[0]
    nblk = (n + 1023) // 1024
    off = torch.empty((max(nblk, 1),), dtype=torch.int32, device=h.device)
    total = torch.empty((1,), dtype=torch.int64, device=h.device)
    call("nrhip_actor_pairs_count", _ptr(h), n, _ptr(off), _ptr(total), _stream())
    P_ = int(total.item())
    si = torch.empty((P_,), dtype=torch.int64, device=h.device)
    ai = torch.empty((P_,), dtype=torch.int32, device=h.device)
    if P_:
        call("nrhip_actor_pairs_write", _ptr(h), n, _ptr(off), _ptr(total), _ptr(si), _ptr(ai), _stream())
    return si, ai


def actor_density(spec: ActorSpec, cand, origins, directions, pixel_area, starts, ends, decoder_weight: Tensor,
                  density: Tensor, ray_flip: Optional[Tensor] = None, return_actor: bool = False):
    """Overwrites density [R,S] (in place) where the sample lies inside an actor box.  -> hit [R,S] bool (return_actor: the
    int32 index of the actor the kernel used -- the highest index containing the sample -- or -1)"""
    r, keep = _c_rays(origins, directions, pixel_area, starts, ends)
    a, keep2 = spec.c_actors()
    cnt, act, w2b, _ = cand
    dw = _chk(decoder_weight.reshape(-1), "decoder_weight")
    dens = _chk(density, "density")
    assert dens.data_ptr() == density.data_ptr(), "density must be contiguous (updated in place)"
    hit = torch.empty((r.n_rays, r.n_samples), dtype=torch.int32, device=dens.device)
    call("nrhip_actor_density", C.byref(a), C.byref(r), _ptr(cnt), _ptr(act), _ptr(w2b), _ptr(dw), dw.numel(),
         _ptr(dens), _ptr(hit), _ptr(None if ray_flip is None else _chk(ray_flip.reshape(-1), "ray_flip")), _stream())
    return hit if return_actor else hit >= 0


@dataclass
class OccGridSpec:
    aabb: Tensor       # [6] min xyz, max xyz (any device; read on the host)
    binaries: Tensor   # [res,res,res] bool / uint8 on the GPU

    def c_grid(self):
        b = self.binaries
        if not b.is_cuda or b.dim() != 3 or b.shape[0] != b.shape[1] or b.shape[1] != b.shape[2]:
            raise ValueError("binaries must be a cubic [res,res,res] GPU tensor")
        b8 = b.to(torch.uint8).contiguous()
        g = _lib.OccGrid()
        for i, v in enumerate(self.aabb.reshape(-1).tolist()):
            g.aabb[i] = v
        g.resolution, g.binaries = b.shape[0], b8.data_ptr()
        return g, b8


def actor_density_splice_fwd(density: Tensor, rows: Tensor, weight: Tensor, sample_idx: Tensor, winner: Tensor):
    """density [N] is overwritten at the hit samples by trunc_exp(rows . weight) of their winning pair -> logit [P]"""
    rows, weight = _chk(rows, "rows"), _chk(weight, "weight")
    idx, win = _chk(sample_idx, "sample_idx", torch.int64), winner.contiguous().view(torch.uint8) if winner.dtype == torch.bool else _chk(winner, "winner", torch.uint8)
    P, la = rows.shape
    if weight.numel() != la or idx.shape[0] != P or win.shape[0] != P or density.dtype != torch.float32 or not density.is_contiguous():
        raise ValueError("actor_density_splice_fwd: shapes")
    logit = torch.empty((P,), device=rows.device, dtype=torch.float32)
    call("nrhip_actor_density_splice_fwd", _ptr(rows), la, _ptr(weight), _ptr(idx), _ptr(win), P, _ptr(density), _ptr(logit),
         _stream())
    return logit


def actor_density_splice_bwd(rows, weight, sample_idx, winner, logit, density_out, grad_out: Tensor):
    """-> grad_density [N] (grad_out, zero at the hit samples), grad_rows [P, la], grad_weight [la]"""
    rows, weight = _chk(rows, "rows"), _chk(weight, "weight")
    idx = _chk(sample_idx, "sample_idx", torch.int64)
    win = winner.contiguous().view(torch.uint8) if winner.dtype == torch.bool else _chk(winner, "winner", torch.uint8)
    g = _chk(grad_out.reshape(-1), "grad_out")
    P, la = rows.shape
    g_dens = g.clone()
    g_rows = torch.empty_like(rows)
    g_w = torch.zeros((la,), device=rows.device, dtype=torch.float32)
    call("nrhip_actor_density_splice_bwd", _ptr(rows), la, _ptr(weight), _ptr(idx), _ptr(win), _ptr(_chk(logit, "logit")),
         _ptr(_chk(density_out.reshape(-1), "density_out")), _ptr(g), P, _ptr(g_dens), _ptr(g_rows), _ptr(g_w), _stream())
    return g_dens, g_rows, g_w


def occgrid_march(grid: OccGridSpec, origins, directions, render_step_size, near_plane=0.0, far_plane=1e10,
                  t_min=None, t_max=None, cone_angle=0.0, t_rand=None, max_candidates=1 << 16):
    """Two-pass packed march: count -> exclusive prefix sum (one host sync for the allocation, like nerfacc) -> write.
    -> (ray_indices int64 [M], t_starts [M], t_ends [M], segments int64 [R+1])"""
    o, d = _chk(origins, "origins"), _chk(directions, "directions")
    R, dev = o.shape[0], o.device
    g, keep = grid.c_grid()
    tmn = None if t_min is None else _chk(t_min.reshape(-1), "t_min")
    tmx = None if t_max is None else _chk(t_max.reshape(-1), "t_max")
    tr = None if t_rand is None else _chk(t_rand.reshape(-1), "t_rand")
    counts = torch.zeros((R,), dtype=torch.int32, device=dev)
    args = (C.byref(g), _ptr(o), _ptr(d), _ptr(tmn), _ptr(tmx), _ptr(tr), R, float(render_step_size), float(near_plane),
            float(far_plane), float(cone_angle), int(max_candidates))
    call("nrhip_occgrid_march", *args, _ptr(counts), _ptr(None), _ptr(None), _ptr(None), _ptr(None), _stream())
    seg = torch.zeros((R + 1,), dtype=torch.int64, device=dev)
    torch.cumsum(counts, 0, out=seg[1:])
    M = int(seg[-1].item()) if R else 0
    ri = torch.empty((M,), dtype=torch.int64, device=dev)
    ts = torch.empty((M,), dtype=torch.float32, device=dev)
    te = torch.empty((M,), dtype=torch.float32, device=dev)
    if M:
        call("nrhip_occgrid_march", *args, _ptr(None), _ptr(seg), _ptr(ri), _ptr(ts), _ptr(te), _stream())
    return ri, ts, te, seg


def packed_visibility_from_alpha(alphas: Tensor, segments: Tensor, early_stop_eps: float, alpha_thre: float) -> Tensor:
    a = _chk(alphas.reshape(-1), "alphas")
    mask = torch.empty((a.shape[0],), dtype=torch.uint8, device=a.device)
    if a.shape[0]:
        call("nrhip_packed_visibility_from_alpha", _ptr(a), _ptr(segments), segments.shape[0] - 1, float(early_stop_eps),
             float(alpha_thre), _ptr(mask), _stream())
    return mask.bool()


def adam_step(param: Tensor, grad: Tensor, exp_avg: Tensor, exp_avg_sq: Tensor, step: int, lr: float, beta1: float = 0.9,
              beta2: float = 0.999, eps: float = 1e-15, weight_decay: float = 0.0, grad_scale: float = 1.0) -> None:
    """torch.optim.Adam / AdamW update of one fp32 tensor, in place (csrc/adam.hip)"""
    for t, n in ((param, "param"), (grad, "grad"), (exp_avg, "exp_avg"), (exp_avg_sq, "exp_avg_sq")):
        if _chk(t, n).data_ptr() != t.data_ptr() or t.shape != param.shape:
            raise ValueError(f"adam_step: {n} must be a contiguous fp32 GPU tensor of the parameter's shape")
    call("nrhip_adam_step", _ptr(param), _ptr(grad), _ptr(exp_avg), _ptr(exp_avg_sq), param.numel(), int(step), float(lr),
         float(beta1), float(beta2), float(eps), float(weight_decay), float(grad_scale), _stream())


def adam_step_many(items, lr: float, beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-15, weight_decay: float = 0.0,
                   grad_scale: float = 1.0) -> None:
    """torch.optim.Adam / AdamW update of MANY tensors in one launch per 24 (csrc/adam.hip).  items: (param fp32, grad fp32 |
    fp16, exp_avg, exp_avg_sq, step, image | None) -- ``image``: the fp16 table whose fp32 master copy ``param`` is; it
    receives the rounded new values in the same pass."""
    items = list(items)
    if not items:
        return
    arr = (_lib.AdamTensor * len(items))()
    for k, (param, grad, m, v, step, image) in enumerate(items):
        for t, n in ((param, "param"), (m, "exp_avg"), (v, "exp_avg_sq")):
            if _chk(t, n).data_ptr() != t.data_ptr() or t.shape != param.shape:
                raise ValueError(f"adam_step_many: {n} must be a contiguous fp32 GPU tensor of the parameter's shape")
        if grad.dtype not in (torch.float32, torch.float16) or not grad.is_contiguous() or not grad.is_cuda or grad.shape != param.shape:
            raise ValueError("adam_step_many: grad must be a contiguous fp32 / fp16 GPU tensor of the parameter's shape")
        if image is not None and (image.dtype != torch.float16 or not image.is_contiguous() or image.shape != param.shape):
            raise ValueError("adam_step_many: image must be a contiguous fp16 tensor of the parameter's shape")
        a = arr[k]
        a.param, a.grad, a.exp_avg, a.exp_avg_sq = param.data_ptr(), grad.data_ptr(), m.data_ptr(), v.data_ptr()
        a.image_fp16 = image.data_ptr() if image is not None else None
        a.n, a.step, a.grad_dtype = param.numel(), int(step), 1 if grad.dtype == torch.float16 else 0
    call("nrhip_adam_step_many", arr, len(items), float(lr), float(beta1), float(beta2), float(eps), float(weight_decay),
         float(grad_scale), _stream())


_ADAM_CTL_BYTES = None


def adam_workspace_floats(n_tensors: int) -> int:
    """size of adam_step_many_dev's workspace for ``n_tensors`` tensors, in fp32 elements"""
    global _ADAM_CTL_BYTES
    if _ADAM_CTL_BYTES is None:
        nb = C.c_int64()
        call("nrhip_adam_step_many_workspace", 1, C.byref(nb))
        _ADAM_CTL_BYTES = int(nb.value)
    return (_ADAM_CTL_BYTES * max(int(n_tensors), 1) + 3) // 4


def adam_workspace(n_tensors: int, device) -> Tensor:
    """device workspace of adam_step_many_dev for up to ``n_tensors`` tensors (the caller keeps it across steps)"""
    return torch.empty((adam_workspace_floats(n_tensors),), device=device, dtype=torch.float32)


def adam_step_many_dev(items, lr, beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-15, weight_decay: float = 0.0,
                       grad_scale: Optional[Tensor] = None, found_inf: Optional[Tensor] = None,
                       workspace: Optional[Tensor] = None, host_grad_scale: float = 1.0) -> None:
    """adam_step_many with everything a step decides ON THE DEVICE (csrc/adam.hip: GradScaler's protocol, graph capture).
    items: (param fp32, grad fp32 | fp16, exp_avg, exp_avg_sq, step = fp32 device scalar holding the count BEFORE this
    update, image | None).  lr: float, or an fp32 device scalar.  grad_scale / found_inf: fp32 device scalars (the
    GradScaler's scale and its found-inf flag): gradients are divided by the scale; a non-zero flag leaves parameters, moments
    and step counts untouched.  No host read."""
    items = list(items)
    if not items:
        return
    dev = items[0][0].device
    arr = (_lib.AdamTensorDev * len(items))()
    for k, (param, grad, m, v, step, image) in enumerate(items):
        for t, n in ((param, "param"), (m, "exp_avg"), (v, "exp_avg_sq")):
            if _chk(t, n).data_ptr() != t.data_ptr() or t.shape != param.shape:
                raise ValueError(f"adam_step_many_dev: {n} must be a contiguous fp32 GPU tensor of the parameter's shape")
        if grad.dtype not in (torch.float32, torch.float16) or not grad.is_contiguous() or not grad.is_cuda or grad.shape != param.shape:
            raise ValueError("adam_step_many_dev: grad must be a contiguous fp32 / fp16 GPU tensor of the parameter's shape")
        if image is not None and (image.dtype != torch.float16 or not image.is_contiguous() or image.shape != param.shape):
            raise ValueError("adam_step_many_dev: image must be a contiguous fp16 tensor of the parameter's shape")
        if not (isinstance(step, Tensor) and step.is_cuda and step.dtype == torch.float32 and step.numel() == 1):
            raise ValueError("adam_step_many_dev: step must be an fp32 GPU scalar")
        a = arr[k]
        a.param, a.grad, a.exp_avg, a.exp_avg_sq = param.data_ptr(), grad.data_ptr(), m.data_ptr(), v.data_ptr()
        a.image_fp16 = image.data_ptr() if image is not None else None
        a.n, a.step, a.grad_dtype = param.numel(), step.data_ptr(), 1 if grad.dtype == torch.float16 else 0

    def scalar(t, what):
        if t is None:
            return None
        if not (isinstance(t, Tensor) and t.is_cuda and t.dtype == torch.float32 and t.numel() == 1):
            raise ValueError(f"adam_step_many_dev: {what} must be an fp32 GPU scalar")
        return t.data_ptr()

    lr_dev = scalar(lr, "lr") if isinstance(lr, Tensor) else None
    if workspace is None:
        workspace = adam_workspace(len(items), dev)
    if workspace.numel() < adam_workspace_floats(len(items)) or not workspace.is_cuda or workspace.dtype != torch.float32:
        raise ValueError("adam_step_many_dev: workspace too small (ops.adam_workspace)")
    call("nrhip_adam_step_many_dev", arr, len(items), 0.0 if lr_dev is not None else float(lr), lr_dev, float(beta1),
         float(beta2), float(eps), float(weight_decay), float(host_grad_scale), scalar(grad_scale, "grad_scale"),
         scalar(found_inf, "found_inf"), workspace.data_ptr(), _stream())


def nonfinite_check(tensors, found_inf: Tensor) -> Tensor:
    """GradScaler's inf check, read-only (csrc/adam.hip: nonfinite_check_kernel): ``found_inf`` (fp32 GPU scalar) becomes 1
    when any element of any of ``tensors`` (contiguous fp32 / fp16 GPU tensors, 16-byte aligned) is inf or NaN; it is never
    cleared here.  Same flag semantics as ``torch._amp_foreach_non_finite_check_and_unscale_`` at a scale of 1, without the
    write-back."""
    tensors = [t for t in tensors if t.numel()]
    if not (found_inf.is_cuda and found_inf.dtype == torch.float32 and found_inf.numel() == 1):
        raise ValueError("nonfinite_check: found_inf must be an fp32 GPU scalar")
    if not tensors:
        return found_inf
    arr = (_lib.CheckTensor * len(tensors))()
    for k, t in enumerate(tensors):
        if (not t.is_cuda or t.dtype not in (torch.float32, torch.float16) or not t.is_contiguous() or t.data_ptr() % 16
                or t.device != found_inf.device):
            raise ValueError("nonfinite_check: tensors must be contiguous, 16-byte aligned fp32 / fp16 tensors on found_inf's GPU")
        arr[k].data, arr[k].n, arr[k].dtype = t.data_ptr(), t.numel(), 1 if t.dtype == torch.float16 else 0
    call("nrhip_nonfinite_check_many", arr, len(tensors), found_inf.data_ptr(), _stream())
    return found_inf


def reload_tuning() -> None:
    """the library reads its NRHIP_* A/B switches once at load; call this after changing one inside a running process"""
    call("nrhip_tuning_reload")


def device_info():
    cus, xcds, hbm = C.c_int32(), C.c_int32(), C.c_int64()
    call("nrhip_device_info", C.byref(cus), C.byref(xcds), C.byref(hbm))
    return {"cus": cus.value, "xcds": xcds.value, "hbm_bytes": hbm.value}


# ------------------------------------------------------------------------------------------------
# losses on the sampler outputs
def interlevel_loss_level(c: Tensor, w: Tensor, cp: Tensor, wp: Tensor, pulse_width: float, need_grad: bool = True):
    """one proposal level of zipnerf_interlevel_loss: -> loss_per_ray [R], d loss_per_ray / d wp [R,Sp] (or None)"""
    c, w, cp, wp = _chk(c, "c"), _chk(w, "w"), _chk(cp, "cp"), _chk(wp, "wp")
    R, sf, sp = w.shape[0], w.shape[1], wp.shape[1]
    if c.shape != (R, sf + 1) or cp.shape != (R, sp + 1) or wp.shape[0] != R:
        raise ValueError(f"interlevel_loss: shapes c {tuple(c.shape)} w {tuple(w.shape)} cp {tuple(cp.shape)} wp {tuple(wp.shape)}")
    loss = torch.empty((R,), device=w.device, dtype=torch.float32)
    g = torch.empty_like(wp) if need_grad else None
    call("nrhip_interlevel_loss", _ptr(c), _ptr(w), sf, _ptr(cp), _ptr(wp), sp, float(pulse_width), R, _ptr(loss), _ptr(g),
         _stream())
    return loss, g


def distortion_loss_rays(c: Tensor, w: Tensor, need_grad: bool = True):
    """lossfun_distortion per ray: -> loss_per_ray [R], d loss_per_ray / d w [R,S] (or None)"""
    c, w = _chk(c, "c"), _chk(w, "w")
    R, s = w.shape
    if c.shape != (R, s + 1):
        raise ValueError(f"distortion_loss: c {tuple(c.shape)} does not match w {tuple(w.shape)}")
    loss = torch.empty((R,), device=w.device, dtype=torch.float32)
    g = torch.empty_like(w) if need_grad else None
    call("nrhip_distortion_loss", _ptr(c), _ptr(w), s, R, _ptr(loss), _ptr(g), _stream())
    return loss, g


# ------------------------------------------------------------------------------------------------
# The training step's glue as kernels (csrc/train_fused.hip): bin EDGES [R,S+1] in, no [R,S,1] views
def _edges(e: Tensor, S: int, name: str = "edges"):
    if not (isinstance(e, Tensor) and e.is_cuda and e.dtype == torch.float32 and e.dim() == 2):
        raise _lib.NeuradHipError(f"{name}: expected a 2-D float32 GPU tensor")
    if e.shape[1] < S + 1:
        raise ValueError(f"{name}: {tuple(e.shape)} holds fewer than S+1 = {S + 1} edges per ray")
    if e.stride(1) != 1:
        e = e.contiguous()
    return e, (e.stride(0) if e.shape[0] > 1 else e.shape[1])


def prop_weights_fwd(edges: Tensor, densities: Tensor, want_depth: bool = True):
    """RaySamples.get_weights + render_depth_simple of one proposal round -> (weights [R,S], depth [R,1] or None)"""
    dens = _chk(densities, "densities")
    R, S = dens.shape
    e, es = _edges(edges, S)
    w = torch.empty_like(dens)
    depth = torch.empty((R, 1), device=dens.device, dtype=torch.float32) if want_depth else None
    call("nrhip_prop_weights_fwd", _ptr(e), es, _ptr(dens), R, S, _ptr(w), _ptr(depth), _stream())
    return w, depth


def prop_weights_bwd(edges: Tensor, densities: Tensor, grad_w: Optional[Tensor], grad_depth: Optional[Tensor]) -> Tensor:
    dens = _chk(densities, "densities")
    R, S = dens.shape
    e, es = _edges(edges, S)
    gw = None if grad_w is None else _chk(grad_w, "grad_w")
    gd = None if grad_depth is None else _chk(grad_depth.reshape(-1), "grad_depth")
    gdens = torch.empty_like(dens)
    call("nrhip_prop_weights_bwd", _ptr(e), es, _ptr(dens), _ptr(gw), _ptr(gd), R, S, _ptr(gdens), _stream())
    return gdens


def sdf_render_fwd(sdf: Tensor, beta: Tensor, beta_min: float, features: Tensor, edges: Tensor, extra_cols: int = 0):
    """SDF head + weights + compositing.  sdf [R,S], beta = the raw learnable parameter (device, 1 element), features
    [R,S,C], edges [R,S+1] (last edge = sky distance).  -> alpha [R,S], weights_ns [R,S-1], out [R,C+extra_cols] (the
    first C columns written), depth [R,1], acc [R,1].  beta = None: the density head (use_sdf = False) -- ``sdf`` is the
    raw geometry output x, sigma = trunc_exp(x), alpha = 1 - exp(-sigma (end - start)) (render_weight_from_density)."""
    sdf, feat = _chk(sdf, "sdf"), _chk(features, "features")
    b = None if beta is None else _chk(beta.reshape(-1), "beta")
    R, S = sdf.shape
    C_ = feat.shape[-1]
    if feat.numel() != R * S * C_ or (b is not None and b.numel() != 1):
        raise ValueError("sdf_render_fwd: features must be [R,S,C], beta one element")
    e, es = _edges(edges, S)
    dev = sdf.device
    alpha = torch.empty_like(sdf)
    w_ns = torch.empty((R, S - 1), device=dev, dtype=torch.float32)
    out = torch.empty((R, C_ + extra_cols), device=dev, dtype=torch.float32)
    depth = torch.empty((R, 1), device=dev, dtype=torch.float32)
    acc = torch.empty((R, 1), device=dev, dtype=torch.float32)
    call("nrhip_sdf_render_fwd", _ptr(sdf), _ptr(b), float(beta_min), _ptr(feat), _ptr(e), es, R, S, C_, _ptr(alpha),
         _ptr(w_ns), _ptr(out), C_ + extra_cols, _ptr(depth), _ptr(acc), _stream())
    return alpha, w_ns, out, depth, acc


def _strided_rows(t: Tensor, name: str):
    """[R, C] float32 GPU view with unit inner stride -> (tensor, row stride); copies only when it has to"""
    if not (t.is_cuda and t.dtype == torch.float32 and t.dim() == 2):
        raise _lib.NeuradHipError(f"{name}: expected a 2-D float32 GPU tensor")
    if t.stride(1) != 1 or (t.shape[0] > 1 and t.stride(0) < t.shape[1]):
        t = t.contiguous()
    return t, (t.stride(0) if t.shape[0] > 1 else t.shape[1])


def sdf_render_bwd(sdf, beta, beta_min, alpha, features, edges, g_out: Tensor, g_depth: Optional[Tensor],
                   g_acc: Optional[Tensor], g_weights_ns: Optional[Tensor]):
    """-> grad_features [R,S,C], grad_sdf [R,S], grad_beta [1] (None for the density head, beta = None).  g_out: [R,C] view
    (row stride free) of the gradient of the composited features."""
    sdf, feat, alpha = _chk(sdf, "sdf"), _chk(features, "features"), _chk(alpha, "alpha")
    b = None if beta is None else _chk(beta.reshape(-1), "beta")
    R, S = sdf.shape
    C_ = feat.shape[-1]
    e, es = _edges(edges, S)
    g, gs = _strided_rows(g_out, "g_out")
    if g.shape != (R, C_):
        raise ValueError(f"sdf_render_bwd: g_out {tuple(g.shape)} != {(R, C_)}")
    if g.data_ptr() % 16 or (gs * 4) % 16:
        g, gs = g.contiguous(), C_
    gd = None if g_depth is None else _chk(g_depth.reshape(-1), "g_depth")
    ga = None if g_acc is None else _chk(g_acc.reshape(-1), "g_acc")
    gw = None if g_weights_ns is None else _chk(g_weights_ns.reshape(R, S - 1), "g_weights_ns")
    dev = sdf.device
    gfeat = torch.empty_like(feat)
    gsdf = torch.empty_like(sdf)
    gbeta = None if b is None else torch.empty((1,), device=dev, dtype=torch.float32)
    need = C.c_int64(0)
    call("nrhip_sdf_render_bwd_workspace", R, C.byref(need))
    ws = torch.empty((max(need.value, 1),), device=dev, dtype=torch.float32)
    call("nrhip_sdf_render_bwd", _ptr(sdf), _ptr(b), float(beta_min), _ptr(alpha), _ptr(feat), _ptr(e), es, _ptr(g), gs,
         _ptr(gd), _ptr(ga), _ptr(gw), R, S, C_, _ptr(gfeat), _ptr(gsdf), _ptr(gbeta), _ptr(ws), _stream())
    return gfeat, gsdf, gbeta


def appearance_fwd(weight: Tensor, sensor_idx: Optional[Tensor], times: Optional[Tensor], duration: float,
                   n_per_sensor: int, temporal: bool, n_rays: int, out: Optional[Tensor] = None) -> Tensor:
    """appearance embedding rows (models/neurad.py:423-441) written into ``out`` ([R,D] view, row stride free)"""
    w = _chk(weight, "weight")
    E, D = w.shape
    s = None if sensor_idx is None else _chk(sensor_idx.reshape(-1), "sensor_idx", torch.int64)
    t = None if times is None else _chk(times.reshape(-1), "times")
    if out is None:
        out = torch.empty((n_rays, D), device=w.device, dtype=torch.float32)
    if not (out.is_cuda and out.dtype == torch.float32 and out.shape == (n_rays, D) and out.stride(1) == 1):
        raise ValueError("appearance_fwd: out must be a float32 [R,D] GPU view with unit inner stride")
    call("nrhip_appearance_fwd", _ptr(w), _ptr(s), _ptr(t), float(duration), int(n_per_sensor), 1 if temporal else 0,
         n_rays, E, D, _ptr(out), out.stride(0) if n_rays > 1 else D, _stream())
    return out


def appearance_bwd(g_out: Tensor, sensor_idx, times, duration: float, n_per_sensor: int, temporal: bool,
                   n_embed: int) -> Tensor:
    g, gs = _strided_rows(g_out, "g_out")
    R, D = g.shape
    s = None if sensor_idx is None else _chk(sensor_idx.reshape(-1), "sensor_idx", torch.int64)
    t = None if times is None else _chk(times.reshape(-1), "times")
    gw = torch.empty((n_embed, D), device=g.device, dtype=torch.float32)  # the entry point zero-fills it
    call("nrhip_appearance_bwd", _ptr(g), gs, _ptr(s), _ptr(t), float(duration), int(n_per_sensor), 1 if temporal else 0,
         R, n_embed, D, _ptr(gw), _stream())
    return gw


def mask_compact(mask: Tensor, n_out: int):
    """rows (int64 [n_out], ascending) where ``mask`` [R] is set + inverse (int32 [R]: slot or -1), without the host sync of
    ``mask.nonzero()`` -- the caller knows n_out (the lidar part of a batch comes with the batch)"""
    m = mask.reshape(-1)
    if not m.is_cuda:
        raise _lib.NeuradHipError("mask_compact: mask is on the CPU (no CPU fallback)")
    m = (m if m.dtype == torch.uint8 else (m.contiguous().view(torch.uint8) if m.dtype == torch.bool else m.ne(0).view(torch.uint8)))
    m = m.contiguous()
    # n_out is the caller's word for mask.sum(); should it be too large, the rows past the real count stay 0 (a valid
    # gather index) instead of uninitialised memory
    rows = torch.zeros((n_out,), device=m.device, dtype=torch.int64)
    inverse = torch.empty((m.shape[0],), device=m.device, dtype=torch.int32)
    call("nrhip_mask_compact", _ptr(m), m.shape[0], _ptr(rows), n_out, _ptr(inverse), _ptr(None), _stream())
    return rows, inverse


def lidar_losses(depths: Sequence[Tensor], lidar_rows: Tensor, distance: Tensor, did_return: Tensor, intensity: Tensor,
                 intensity_target: Tensor, ray_drop_logits: Tensor, non_return_distance: float, non_return_mult: float,
                 quantile: float):
    """-> metrics [2 + n_levels] (depth_loss, intensity_loss, ray_drop_loss, depth_loss_0, ...), and what the backward
    needs (per-ray gradients, scratch with the errors / threshold / counts, did_return as uint8)"""
    nl = len(depths)
    ds = [_chk(d.reshape(-1), f"depth[{i}]") for i, d in enumerate(depths)]
    rows = _chk(lidar_rows, "lidar_rows", torch.int64)
    n = rows.shape[0]
    dist, inten = _chk(distance.reshape(-1), "distance"), _chk(intensity.reshape(-1), "intensity")
    tgt, lg = _chk(intensity_target.reshape(-1), "intensity_target"), _chk(ray_drop_logits.reshape(-1), "ray_drop_logits")
    ret = did_return.reshape(-1).contiguous()
    ret = ret.view(torch.uint8) if ret.dtype == torch.bool else ret.to(torch.uint8)
    for v, nm in ((dist, "distance"), (inten, "intensity"), (tgt, "intensity_target"), (lg, "ray_drop_logits"), (ret, "did_return")):
        if v.shape[0] != n:
            raise ValueError(f"lidar_losses: {nm} has {v.shape[0]} rows, expected {n}")
    dev = rows.device
    metrics = torch.empty((2 + nl,), device=dev, dtype=torch.float32)
    unit = torch.empty((nl + 2, n), device=dev, dtype=torch.float32)
    need = C.c_int64(0)
    call("nrhip_lidar_losses_workspace", n, C.byref(need))
    scratch = torch.empty((need.value,), device=dev, dtype=torch.float32)
    pd = (C.c_void_p * nl)(*[d.data_ptr() for d in ds])
    call("nrhip_lidar_losses", C.cast(pd, C.POINTER(C.c_void_p)), nl, _ptr(rows), _ptr(dist), _ptr(ret), _ptr(inten),
         _ptr(tgt), _ptr(lg), n, float(non_return_distance), float(non_return_mult), float(quantile), _ptr(metrics),
         _ptr(unit), _ptr(scratch), _stream())
    return metrics, (unit, scratch, ret)


def lidar_losses_bwd(saved, inverse: Tensor, upstream: Tensor, n_levels: int, n_rays: int, need_depth: Sequence[bool],
                     need_intensity: bool = True, need_logits: bool = True):
    """saved = what lidar_losses returned beside the metrics.  -> ([grad depth [R,1] or None per level], grad intensity
    [n,1] or None, grad logits [n,1] or None)"""
    unit, scratch, ret = saved
    unit, inv, up = _chk(unit, "unit"), _chk(inverse, "inverse", torch.int32), _chk(upstream.reshape(-1), "upstream")
    n, dev = unit.shape[1], unit.device
    gds = [torch.empty((n_rays, 1), device=dev, dtype=torch.float32) if nd else None for nd in need_depth]
    gi = torch.empty((n, 1), device=dev, dtype=torch.float32) if need_intensity else None
    gl = torch.empty((n, 1), device=dev, dtype=torch.float32) if need_logits else None
    pg = (C.c_void_p * n_levels)(*[(0 if g is None else g.data_ptr()) for g in gds])
    call("nrhip_lidar_losses_bwd", _ptr(unit), _ptr(scratch), _ptr(ret), _ptr(inv), _ptr(up), n_levels, n_rays, n,
         C.cast(pg, C.POINTER(C.c_void_p)), _ptr(gi), _ptr(gl), _stream())
    return gds, gi, gl


# ---- the level-sparse gradient exchange's device side (csrc/grad_rows.hip; parallel/data_parallel.py) ----------
def grad_rows_count(grad: Tensor, n_levels: int):
    """grad [n_levels * T, F] fp32 -> (level_counts [n_levels] int64 = non-zero rows per level, block_offsets [n_levels, nblk]
    uint32 as int32 storage: the per-block exclusive prefix ``grad_rows_compact`` needs)"""
    g = _chk(grad, "grad")
    T, F = g.shape[0] // n_levels, g.shape[1]
    nblk = (T + _lib.GRAD_ROWS_PER_BLOCK - 1) // _lib.GRAD_ROWS_PER_BLOCK
    blocks = torch.empty((n_levels, nblk), dtype=torch.int32, device=g.device)
    counts = torch.empty((n_levels,), dtype=torch.int64, device=g.device)
    call("nrhip_grad_rows_count", _ptr(g), n_levels, T, F, _ptr(blocks), _ptr(counts), _stream())
    return counts, blocks


_MAX_LIST_LEVELS = 32  # kMaxListLevels of csrc/grad_rows.hip


def _list_args(levels: Sequence[int], caps: Sequence[int]):
    n = len(levels)
    return (C.c_int32 * n)(*levels), (C.c_int64 * n)(*caps), n


def grad_rows_compact(grad: Tensor, n_levels: int, block_offsets: Tensor, levels: Sequence[int], caps: Sequence[int],
                      scale: float = 1.0):
    """the levels ``levels`` of grad as ordered (row, values) lists, level i padded to caps[i] entries with row -1 / zeros
    -> (rows [sum caps] int32, vals [sum caps, F] fp32 = grad * scale)"""
    g = _chk(grad, "grad")
    T, F = g.shape[0] // n_levels, g.shape[1]
    total = int(sum(caps))
    rows = torch.full((total,), -1, dtype=torch.int32, device=g.device)
    vals = torch.zeros((total, F), dtype=torch.float32, device=g.device)
    bo = _chk(block_offsets, "block_offsets", torch.int32)
    off = 0
    for k in range(0, len(levels), _MAX_LIST_LEVELS):  # the kernel takes at most 32 list levels per launch
        lvk, cpk = list(levels[k:k + _MAX_LIST_LEVELS]), list(caps[k:k + _MAX_LIST_LEVELS])
        tk = int(sum(cpk))
        if tk:
            lv, cp, n = _list_args(lvk, cpk)
            call("nrhip_grad_rows_compact", _ptr(g), n_levels, T, F, _ptr(bo), lv, cp, n, float(scale), _ptr(rows[off:]),
                 _ptr(vals[off:]), _stream())
        off += tk
    return rows, vals


def grad_rows_apply(grad: Tensor, n_levels: int, levels: Sequence[int], caps: Sequence[int], rows: Tensor,
                    vals: Optional[Tensor], add: bool) -> None:
    """in place on grad: rows of the list -> 0 (add=False) or += vals (add=True); entries with row -1 are padding"""
    if not grad.is_contiguous():
        raise ValueError("grad_rows_apply: contiguous gradient only (it is updated in place)")
    g = _chk(grad, "grad")
    T, F = g.shape[0] // n_levels, g.shape[1]
    if int(sum(caps)) == 0:
        return
    rows = _chk(rows, "rows", torch.int32)
    vals = None if vals is None else _chk(vals, "vals")
    off = 0
    for k in range(0, len(levels), _MAX_LIST_LEVELS):
        lvk, cpk = list(levels[k:k + _MAX_LIST_LEVELS]), list(caps[k:k + _MAX_LIST_LEVELS])
        tk = int(sum(cpk))
        if tk:
            lv, cp, n = _list_args(lvk, cpk)
            call("nrhip_grad_rows_apply", _ptr(g), n_levels, T, F, lv, cp, n, _ptr(rows[off:]),
                 _ptr(None if vals is None else vals[off:]), 1 if add else 0, _stream())
        off += tk
