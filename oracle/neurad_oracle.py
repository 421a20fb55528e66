"""CPU restatement (numpy, fp32) of NeuRAD's volumetric ray-marching hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package (``neurad_studio_amd``)
imports this module: it is the *checker* used by ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``.

Every function restates one row of SURVEY.md §8(a) and cites the reference
``file:line`` (paths relative to the neurad-studio tree) it follows.  The parity
target is the reference's ``implementation="torch"`` branch (fp32), NOT tiny-cuda-nn.

Pinning: ``tests/test_oracle_golden.py`` checks every function here against golden
vectors produced by importing the reference itself (``oracle/make_golden.py`` ->
``tests/golden/*.npz``).  Compositing (nerfacc 0.5.2, un-vendored, absent from the
reference tree, and replaced by a 0.5 placeholder on CPU -- models/neurad.py:713-715)
has no reference output to pin against: for those two functions parity is pinned only
against the in-repo torch equivalents ``RaySamples.get_weights`` (cameras/rays.py:188-210)
and is otherwise "parity unpinned" (see DESIGN.md).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

f32 = np.float32
PRIME_Y = np.int64(2654435761)  # field_components/encodings.py:419
PRIME_Z = np.int64(805459861)


# --------------------------------------------------------------------------------------
# H1  HashEncoding (field_components/encodings.py:326-471)
# --------------------------------------------------------------------------------------
def hash_scalings(num_levels: int, min_res: int, max_res: int) -> np.ndarray:
    """``scalings_l = floor(min_res * g**l)``, ``g = exp((ln max - ln min)/(L-1))``.

    encodings.py:347-350.  The reference evaluates ``growth_factor ** levels`` as
    (numpy float64 scalar) ** (torch int64 tensor) -> torch promotes to the default
    dtype float32, multiplies by ``min_res`` and floors, all in fp32.
    """
    if num_levels > 1:
        growth = np.exp((np.log(max_res) - np.log(min_res)) / (num_levels - 1))
    else:
        growth = 1.0
    levels = np.arange(num_levels)
    # torch: (python float) ** (int64 tensor) -> float32 tensor computed by powf(float(g), float(l))
    g = np.power(f32(growth), levels.astype(f32), dtype=f32)
    return np.floor(f32(min_res) * g).astype(f32)


def hash_indices(corner: np.ndarray, table_size: int, level_offset: np.ndarray) -> np.ndarray:
    """``hash_fn`` (encodings.py:408-423): int32 corners * int64 primes, xor, mod T, + l*T."""
    c = corner.astype(np.int64)
    x = c[..., 0] ^ (c[..., 1] * PRIME_Y) ^ (c[..., 2] * PRIME_Z)
    x = np.mod(x, np.int64(table_size))
    return x + level_offset


def hashgrid_corner_indices(x: np.ndarray, scalings: np.ndarray, table_size: int):
    """All 8 corner indices + interpolation offsets of ``pytorch_fwd`` (encodings.py:425-444).

    Returns (idx [N,L,8] int64 in the reference's corner order 0..7, offset [N,L,3] fp32).
    """
    x = np.asarray(x, f32)
    L = scalings.shape[0]
    scaled = x[:, None, :] * scalings.reshape(L, 1).astype(f32)  # [N,L,3]
    c = np.ceil(scaled).astype(np.int32)
    f = np.floor(scaled).astype(np.int32)
    offset = scaled - f.astype(f32)
    lo = (np.arange(L, dtype=np.int64) * table_size)[None, :]

    def pick(sx, sy, sz):
        return np.stack([sx[..., 0], sy[..., 1], sz[..., 2]], axis=-1)

    combos = [  # encodings.py:437-444
        (c, c, c), (c, f, c), (f, f, c), (f, c, c), (c, c, f), (c, f, f), (f, f, f), (f, c, f),
    ]
    idx = np.stack([hash_indices(pick(*cb), table_size, lo) for cb in combos], axis=-1)
    return idx, offset


def hashgrid_fwd(x: np.ndarray, table: np.ndarray, scalings: np.ndarray, table_size: int) -> np.ndarray:
    """``HashEncoding.pytorch_fwd`` (encodings.py:425-466).  x [N,3] in [0,1] -> [N, L*F]."""
    idx, o = hashgrid_corner_indices(x, scalings, table_size)
    t = np.asarray(table, f32)
    fc = [t[idx[..., k]] for k in range(8)]  # each [N,L,F]
    ox, oy, oz = o[..., 0:1], o[..., 1:2], o[..., 2:3]
    one = f32(1.0)
    f03 = fc[0] * ox + fc[3] * (one - ox)
    f12 = fc[1] * ox + fc[2] * (one - ox)
    f56 = fc[5] * ox + fc[6] * (one - ox)
    f47 = fc[4] * ox + fc[7] * (one - ox)
    f0312 = f03 * oy + f12 * (one - oy)
    f4756 = f47 * oy + f56 * (one - oy)
    enc = f0312 * oz + f4756 * (one - oz)
    return enc.reshape(x.shape[0], -1).astype(f32)


def hashgrid_bwd(x, grad_out, scalings, table_size, n_rows, n_feat) -> np.ndarray:
    """dL/d(hash_table) of :func:`hashgrid_fwd` (autograd of encodings.py:446-464): scatter-add of
    the 8 trilinear corner weights times the upstream gradient.  Accumulates in float64."""
    idx, o = hashgrid_corner_indices(x, scalings, table_size)
    N, L = idx.shape[:2]
    g = np.asarray(grad_out, np.float64).reshape(N, L, n_feat)
    ox, oy, oz = (o[..., k].astype(np.float64) for k in range(3))
    wx = {"c": ox, "f": 1 - ox}
    wy = {"c": oy, "f": 1 - oy}
    wz = {"c": oz, "f": 1 - oz}
    order = ["ccc", "cfc", "ffc", "fcc", "ccf", "cff", "fff", "fcf"]
    out = np.zeros((n_rows, n_feat), np.float64)
    for k, s in enumerate(order):
        w = wx[s[0]] * wy[s[1]] * wz[s[2]]
        np.add.at(out, idx[..., k].reshape(-1), (w[..., None] * g).reshape(-1, n_feat))
    return out.astype(f32)


# --------------------------------------------------------------------------------------
# F3  SHEncoding (encodings.py:797-805 -> utils/math.py:31-94), levels=4
# --------------------------------------------------------------------------------------
def sh_deg4(d: np.ndarray) -> np.ndarray:
    """16 real SH components of ``d`` -- NB the torch path feeds ``(dir+1)/2`` unchanged
    (fields/base_field.py:136-142, neurad_field.py:140)."""
    d = np.asarray(d, f32)
    x, y, z = d[..., 0], d[..., 1], d[..., 2]
    xx, yy, zz = x * x, y * y, z * z
    c = np.zeros(d.shape[:-1] + (16,), f32)
    c[..., 0] = 0.28209479177387814
    c[..., 1] = f32(0.4886025119029199) * y
    c[..., 2] = f32(0.4886025119029199) * z
    c[..., 3] = f32(0.4886025119029199) * x
    c[..., 4] = f32(1.0925484305920792) * x * y
    c[..., 5] = f32(1.0925484305920792) * y * z
    c[..., 6] = f32(0.9461746957575601) * zz - f32(0.31539156525251999)
    c[..., 7] = f32(1.0925484305920792) * x * z
    c[..., 8] = f32(0.5462742152960396) * (xx - yy)
    c[..., 9] = f32(0.5900435899266435) * y * (f32(3) * xx - yy)
    c[..., 10] = f32(2.890611442640554) * x * y * z
    c[..., 11] = f32(0.4570457994644658) * y * (f32(5) * zz - f32(1))
    c[..., 12] = f32(0.3731763325901154) * z * (f32(5) * zz - f32(3))
    c[..., 13] = f32(0.4570457994644658) * x * (f32(5) * zz - f32(1))
    c[..., 14] = f32(1.445305721320277) * z * (xx - yy)
    c[..., 15] = f32(0.5900435899266435) * x * (xx - f32(3) * yy)
    return c


# --------------------------------------------------------------------------------------
# F2  MLP.pytorch_fwd (field_components/mlp.py:159-178): Linear(+bias)+ReLU ... Linear
# --------------------------------------------------------------------------------------
def mlp_fwd(x: np.ndarray, weights: Sequence[np.ndarray], biases: Sequence[Optional[np.ndarray]],
            return_hidden: bool = False):
    """weights[i] is ``[out,in]`` like ``nn.Linear.weight``; ReLU between layers, none at the end."""
    h = np.asarray(x, f32)
    hidden = [h]
    n = len(weights)
    for i, (w, b) in enumerate(zip(weights, biases)):
        h = h @ np.asarray(w, f32).T
        if b is not None:
            h = h + np.asarray(b, f32)
        if i < n - 1:
            h = np.maximum(h, f32(0))
        hidden.append(h)
    return (h, hidden) if return_hidden else h


def mlp_bwd(hidden: List[np.ndarray], weights, grad_out):
    """Backward of :func:`mlp_fwd` -> (dx, [dW], [db]) in float64 accumulate."""
    g = np.asarray(grad_out, np.float64)
    n = len(weights)
    dWs, dbs = [None] * n, [None] * n
    for i in reversed(range(n)):
        if i < n - 1:
            g = g * (hidden[i + 1] > 0)
        dWs[i] = (g.T @ hidden[i].astype(np.float64)).astype(f32)
        dbs[i] = g.sum(0).astype(f32)
        g = g @ np.asarray(weights[i], np.float64)
    return g.astype(f32), dWs, dbs


# --------------------------------------------------------------------------------------
# H2  Frustums.get_fast_isotropic_gaussian(M=1) (cameras/rays.py:109-124)
# --------------------------------------------------------------------------------------
def fast_isotropic_gaussian(origins, directions, pixel_area, starts, ends):
    """origins/directions [R,3], pixel_area [R], starts/ends [R,S] -> mean [R,S,3], std [R,S]."""
    o, d = np.asarray(origins, f32), np.asarray(directions, f32)
    s, e = np.asarray(starts, f32), np.asarray(ends, f32)
    dist = (e - s) / f32(2)
    t = s + f32(1) * dist
    mean = o[:, None, :] + d[:, None, :] * t[..., None]
    area = np.asarray(pixel_area, f32).reshape(-1, 1) * np.power(t, f32(2))
    std = np.power(area * dist, f32(1 / 3), dtype=f32)
    return mean.astype(f32), std.astype(f32)


# --------------------------------------------------------------------------------------
# H3  ScaledSceneContraction(order=inf) on GaussiansStd (spatial_distortions.py:103-141)
# --------------------------------------------------------------------------------------
def contract_gaussian(mean, std, scale: float):
    """-> positions in [0,1]^3 and contracted std."""
    m = np.asarray(mean, f32) / f32(scale)
    s = np.asarray(std, f32) / f32(scale)
    mag = np.max(np.abs(m), axis=-1, keepdims=True)
    mask = mag < 1
    cm = np.maximum(mag, f32(1))
    m2 = np.where(mask, m, (f32(2) - (f32(1) / cm)) * (m / cm))
    sc = (np.power(f32(2) * cm - f32(1), f32(1 / 3), dtype=f32) / cm) ** 2
    s2 = np.where(mask[..., 0], s, s * sc[..., 0])
    m2 = (m2 + f32(2)) / f32(4)
    s2 = s2 / f32(4)
    return m2.astype(f32), s2.astype(f32)


# --------------------------------------------------------------------------------------
# H4  NeuRADHashEncoding._rescale_grid_features (field_components/neurad_encoding.py:297-304), M=1
# --------------------------------------------------------------------------------------
def rescale_grid_features(feat, std, scalings, n_feat: int):
    """feat [N, L*F], std [N] -> feat * 1/max(1, 2*scalings_l*std)."""
    L = scalings.shape[0]
    w = f32(1) / np.maximum(scalings[None, :].astype(f32) * f32(2) * np.asarray(std, f32).reshape(-1, 1), f32(1))
    return (feat.reshape(-1, L, n_feat) * w[..., None]).reshape(-1, L * n_feat).astype(f32)


# --------------------------------------------------------------------------------------
# Parameter containers (state_dict layout of SURVEY.md Appendix B)
# --------------------------------------------------------------------------------------
@dataclass
class GridParams:
    table: np.ndarray  # [L*T, F]
    num_levels: int
    min_res: int
    max_res: int
    log2_hashmap_size: int

    @property
    def table_size(self):
        return 1 << self.log2_hashmap_size

    @property
    def n_feat(self):
        return self.table.shape[1]

    @property
    def scalings(self):
        return hash_scalings(self.num_levels, self.min_res, self.max_res)


@dataclass
class FieldParams:
    """NeuRADField (fields/neurad_field.py:78-152) parameters, no actors."""

    grid: GridParams
    static_scale: float
    geo_w: List[np.ndarray]
    geo_b: List[np.ndarray]
    feat_w: List[np.ndarray]
    feat_b: List[np.ndarray]
    beta: float = 20.0
    beta_min: float = 1e-4
    use_sdf: bool = True


@dataclass
class ProposalParams:
    """NeuRADProposalField (fields/neurad_field.py:182-216)."""

    grid: GridParams
    static_scale: float
    decoder_w: np.ndarray  # [1, L*F], no bias


def encode_static(grid: GridParams, static_scale, origins, directions, pixel_area, starts, ends, num_multisamples: int = 1):
    """H2 -> H3 -> H1 -> H4 (neurad_encoding.py:164-169,265-268).  -> [R*S, L*F]
    num_multisamples = M > 1 (NeuRADFieldConfig.num_multisamples, fields/neurad_field.py:67,134): the frustum is probed at
    t_k = start + k (end - start) / (M + 1), k = 1..M, each with std_k = (area t_k^2 (end - start) / (M + 1))^(1/3)
    (cameras/rays.py:109-124), and the RESCALED features are averaged over k (neurad_encoding.py:302: mean(dim=-3)).  Sub-sample
    k is exactly the M = 1 gaussian of the interval (t_k - step, t_k + step)."""
    if num_multisamples > 1:
        st, en = np.asarray(starts, f32), np.asarray(ends, f32)
        step = ((en - st) / f32(num_multisamples + 1)).astype(f32)
        acc = None
        for k in range(1, num_multisamples + 1):
            tk = (st + f32(k) * step).astype(f32)
            fk = encode_static(grid, static_scale, origins, directions, pixel_area, (tk - step).astype(f32), (tk + step).astype(f32))
            acc = fk if acc is None else acc + fk
        return (acc / f32(num_multisamples)).astype(f32)
    mean, std = fast_isotropic_gaussian(origins, directions, pixel_area, starts, ends)
    pos, cstd = contract_gaussian(mean, std, static_scale)
    feat = hashgrid_fwd(pos.reshape(-1, 3), grid.table, grid.scalings, grid.table_size)
    return rescale_grid_features(feat, cstd.reshape(-1), grid.scalings, grid.n_feat)


def encode_static_ray_grads(grid: GridParams, static_scale, origins, directions, pixel_area, starts, ends, grad_enc):
    """dL/d(origins), dL/d(directions) [R,3] of :func:`encode_static` (M = 1) given dL/d(rescaled features) [R*S, L*F]:
    what autograd does for a camera optimizer that moves the rays (cameras/camera_optimizers.py:173-182).  The chain, every
    link the derivative of the reference line it names (sample midpoints t are constants: bins are detached,
    ray_samplers.py:363-364):
      enc_l = w_l(s) * lerp_l(x)            neurad_encoding.py:297-304, w_l = 1 / max(1, 2 scal_l s)
      lerp_l: trilinear in offset = x scal_l - floor(x scal_l)          encodings.py:425-464 (floor / ceil carry no gradient)
      x = (c + 2) / 4, s = std' / 4;  |u|_inf >= 1: c = (2 - 1/m) u / m, std' = (std/scale) ((2m - 1)^(1/3) / m)^2, m = |u|_inf
                                                                        spatial_distortions.py:126-141,  u = mean / scale
      mean = o + d t                                                    cameras/rays.py:119
    Primal quantities (cells, offsets, masks) in fp32 exactly as the forward restatement, derivative arithmetic in fp64."""
    f64 = np.float64
    mean, std = fast_isotropic_gaussian(origins, directions, pixel_area, starts, ends)
    R, S = np.asarray(starts).shape
    pos, cstd = contract_gaussian(mean, std, static_scale)
    x = pos.reshape(-1, 3)
    N = x.shape[0]
    sc = grid.scalings.astype(f32)
    L, F = sc.shape[0], grid.n_feat
    idx, off = hashgrid_corner_indices(x, sc, grid.table_size)
    t = np.asarray(grid.table, f64)
    fc = [t[idx[..., k]] for k in range(8)]  # [N,L,F]; corner order 0 ccc 1 cfc 2 ffc 3 fcc 4 ccf 5 cff 6 fff 7 fcf
    ox, oy, oz = (off[..., k:k + 1].astype(f64) for k in range(3))
    mx, my, mz = 1 - ox, 1 - oy, 1 - oz
    f03, f12 = fc[0] * ox + fc[3] * mx, fc[1] * ox + fc[2] * mx
    f56, f47 = fc[5] * ox + fc[6] * mx, fc[4] * ox + fc[7] * mx
    val = (f03 * oy + f12 * my) * oz + (f47 * oy + f56 * my) * mz
    d03, d12, d56, d47 = fc[0] - fc[3], fc[1] - fc[2], fc[5] - fc[6], fc[4] - fc[7]
    dvx = (d03 * oy + d12 * my) * oz + (d47 * oy + d56 * my) * mz
    dvy = (f03 - f12) * oz + (f47 - f56) * mz
    dvz = (f03 * oy + f12 * my) - (f47 * oy + f56 * my)
    g = np.asarray(grad_enc, f64).reshape(N, L, F)
    s = cstd.reshape(-1, 1).astype(f64)
    a = sc[None, :].astype(f64) * 2.0 * s                     # [N,L]
    w = 1.0 / np.maximum(a, 1.0)
    dw = np.where(a > 1.0, -2.0 * sc[None, :].astype(f64) * w * w, 0.0)
    scw = (sc[None, :].astype(f64) * w)[..., None]
    gx = np.stack([(g * dvx * scw).sum((1, 2)), (g * dvy * scw).sum((1, 2)), (g * dvz * scw).sum((1, 2))], -1)  # dL/dx01
    gs = ((g * val).sum(-1) * dw).sum(-1)                    # dL/d cstd
    # contraction backward
    u = (mean.reshape(-1, 3) / f32(static_scale)).astype(f32).astype(f64)
    sd = (std.reshape(-1) / f32(static_scale)).astype(f32).astype(f64)
    au = np.abs(u)
    kmax = np.argmax(au, -1)
    m = au[np.arange(N), kmax]
    gc = gx / 4.0
    outside = ~(m < 1.0)
    mm = np.where(outside, m, 1.0)
    k = 2.0 / mm - 1.0 / mm**2
    dk = -2.0 / mm**2 + 2.0 / mm**3
    cr = np.cbrt(2.0 * mm - 1.0)
    dq = 2.0 * (cr / mm) * ((2.0 / 3.0) / (cr * cr * mm) - cr / mm**2)      # d/dm of ((2m-1)^(1/3)/m)^2
    g_mag = (gc * u).sum(-1) * dk + (gs / 4.0) * sd * dq
    gu = np.where(outside[:, None], gc * k[:, None], gc)
    gu[np.arange(N), kmax] += np.where(outside, g_mag * np.sign(u[np.arange(N), kmax]), 0.0)
    gmean = (gu / f64(static_scale)).reshape(R, S, 3)
    st, en = np.asarray(starts, f32), np.asarray(ends, f32)
    tm = (st + f32(1) * ((en - st) / f32(2))).astype(f64)
    return gmean.sum(1).astype(f32), (gmean * tm[..., None]).sum(1).astype(f32)


def proposal_density_ray_grads(p: "ProposalParams", origins, directions, pixel_area, starts, ends, grad_density):
    """dL/d(origins, directions) of :func:`proposal_density`: trunc_exp's backward g * exp(clamp(x, -15, 15))
    (field_components/activations.py:27-41) -> the decoder row -> :func:`encode_static_ray_grads`."""
    enc = encode_static(p.grid, p.static_scale, origins, directions, pixel_area, starts, ends)
    logit = enc.astype(np.float64) @ np.asarray(p.decoder_w, np.float64).T           # [N,1]
    gl = np.asarray(grad_density, np.float64).reshape(-1, 1) * np.exp(np.clip(logit, -15.0, 15.0))
    genc = gl * np.asarray(p.decoder_w, np.float64).reshape(1, -1)
    return encode_static_ray_grads(p.grid, p.static_scale, origins, directions, pixel_area, starts, ends, genc)


def sigmoid(x):
    x = np.asarray(x, f32)
    return (f32(1) / (f32(1) + np.exp(-x))).astype(f32)


# --------------------------------------------------------------------------------------
# F1/F4  NeuRADField.forward (fields/neurad_field.py:128-152), SigmoidDensity (model_components/utils.py:21-41)
# --------------------------------------------------------------------------------------
def field_fwd(p: FieldParams, origins, directions, pixel_area, starts, ends, num_multisamples: int = 1) -> Dict[str, np.ndarray]:
    """-> {"feature" [R,S,C], "sdf" [R,S], "alpha" [R,S]}  (or "density" when use_sdf=False)."""
    R, S = np.asarray(starts).shape
    enc = encode_static(p.grid, p.static_scale, origins, directions, pixel_area, starts, ends, num_multisamples)
    geo = mlp_fwd(enc, p.geo_w, p.geo_b)
    geo_out, geo_emb = geo[:, :1], geo[:, 1:]
    d01 = (np.asarray(directions, f32) + f32(1)) / f32(2)  # base_field.py:136-142
    sh = sh_deg4(np.broadcast_to(d01[:, None, :], (R, S, 3)).reshape(-1, 3))
    feat = geo_emb + mlp_fwd(np.concatenate([geo_emb, sh], -1), p.feat_w, p.feat_b)
    out = {"feature": feat.reshape(R, S, -1).astype(f32)}
    if p.use_sdf:
        beta = f32(abs(p.beta) + p.beta_min)
        out["sdf"] = geo_out.reshape(R, S)
        out["alpha"] = sigmoid(-geo_out.reshape(R, S) * beta)
    else:
        out["density"] = np.exp(geo_out.reshape(R, S)).astype(f32)  # trunc_exp fwd, activations.py:33-35
    return out


# --------------------------------------------------------------------------------------
# S2  NeuRADProposalField.get_density (fields/neurad_field.py:208-213)
# --------------------------------------------------------------------------------------
def proposal_density(p: ProposalParams, origins, directions, pixel_area, starts, ends) -> np.ndarray:
    R, S = np.asarray(starts).shape
    enc = encode_static(p.grid, p.static_scale, origins, directions, pixel_area, starts, ends)
    return np.exp(enc @ np.asarray(p.decoder_w, f32).T).reshape(R, S).astype(f32)


# --------------------------------------------------------------------------------------
# S3  RaySamples.get_weights (cameras/rays.py:188-210)
# --------------------------------------------------------------------------------------
def weights_from_density(deltas, densities) -> np.ndarray:
    dd = np.asarray(deltas, f32) * np.asarray(densities, f32)
    alphas = f32(1) - np.exp(-dd)
    # torch.cumsum on CPU accumulates fp32 in float64 (at::acc_type<float,false>) and rounds each output
    cs = np.cumsum(dd[..., :-1].astype(np.float64), axis=-1).astype(f32)
    trans = np.exp(-np.concatenate([np.zeros_like(dd[..., :1]), cs], -1))
    return np.nan_to_num(alphas * trans).astype(f32)


# --------------------------------------------------------------------------------------
# C1  nerfacc 0.5.2 dense-mode restatement (un-vendored; call sites models/neurad.py:716-723)
# --------------------------------------------------------------------------------------
def render_weight_from_alpha(alphas):
    """``T_i = prod_{j<i}(1-a_j)``, ``w = T a``  (exclusive cumprod).  -> (weights, trans)"""
    a = np.asarray(alphas, f32)
    om = (f32(1) - a).astype(np.float64)
    trans = np.concatenate([np.ones_like(om[..., :1]), np.cumprod(om[..., :-1], axis=-1)], -1).astype(f32)
    return (trans * a).astype(f32), trans


def render_weight_from_density(t_starts, t_ends, sigmas):
    """``T_i = exp(-sum_{j<i} s_j d_j)``, ``a = 1-exp(-s d)``.  -> (weights, trans, alphas)"""
    sd = np.asarray(sigmas, f32) * (np.asarray(t_ends, f32) - np.asarray(t_starts, f32))
    alphas = f32(1) - np.exp(-sd)
    cs = np.cumsum(sd[..., :-1].astype(np.float64), axis=-1).astype(f32)
    trans = np.exp(-np.concatenate([np.zeros_like(sd[..., :1]), cs], -1)).astype(f32)
    return (trans * alphas).astype(f32), trans, alphas.astype(f32)


def accumulate_along_rays(weights, values=None):
    """dense mode (ray_indices=None): ``sum_S w * v`` -> [R,C] (C=1 when values is None)."""
    w = np.asarray(weights, f32)
    if values is None:
        return w.astype(np.float64).sum(-1, keepdims=True).astype(f32)
    return (w[..., None].astype(np.float64) * np.asarray(values, np.float64)).sum(-2).astype(f32)


# --------------------------------------------------------------------------------------
# C2  get_nff_outputs compositing (models/neurad.py:377-395,727-734; renderers.py:59-90,322-350)
# --------------------------------------------------------------------------------------
def composite(weights, features, starts, ends):
    """weights [R,S] (from C1), features [R,S,C] -> features [R,C], depth [R,1], accumulation [R,1].

    acc = sum(w); the residual 1-acc goes onto the last (sky) sample; features use all S samples;
    depth drops the sky sample and is NOT normalised (render_depth_simple).
    """
    w = np.asarray(weights, f32)
    acc = w.astype(np.float64).sum(-1, keepdims=True).astype(f32)
    w2 = np.concatenate([w[..., :-1], w[..., -1:] + f32(1) - acc], -1)
    feat = (w2[..., None].astype(np.float64) * np.asarray(features, np.float64)).sum(-2).astype(f32)
    steps = (np.asarray(starts, f32) + np.asarray(ends, f32)) / f32(2)
    depth = (w2[..., :-1].astype(np.float64) * steps[..., :-1]).sum(-1, keepdims=True).astype(f32)
    return feat, depth, acc


def depth_expected(weights, starts, ends):
    """DepthRenderer("expected") dense branch (model_components/renderers.py:398-416)."""
    w = np.asarray(weights, f32)
    steps = (np.asarray(starts, f32) + np.asarray(ends, f32)) / f32(2)
    d = (w * steps).sum(-1, keepdims=True) / (w.sum(-1, keepdims=True) + f32(1e-10))
    return np.clip(d, steps.min(), steps.max()).astype(f32)


# --------------------------------------------------------------------------------------
# S1  SpacedSampler / PowerSampler (ray_samplers.py:80-132,838-852; utils/math.py:541-579)
# --------------------------------------------------------------------------------------
def _pow_like_aten(base, e: float):
    """``tensor ** python_float``: ATen evaluates the exponent -1 (NeuRAD's lambda and its inverse) as the reciprocal --
    torch.pow(x, -1.0) == 1 / x bit for bit -- where numpy's power is an ulp off on a fifth of the inputs"""
    if e == -1.0:
        return (f32(1) / base).astype(f32)
    return np.power(base, f32(e), dtype=f32)


def power_fn(x, lam: float):
    x = np.asarray(x, f32)
    lam_1 = abs(lam - 1)
    return (f32(lam_1 / lam) * (_pow_like_aten(x / f32(lam_1) + f32(1), lam) - f32(1))).astype(f32)


def inv_power_fn(x, lam: float, eps: float = 1e-10):
    x = np.asarray(x, f32)
    lam_1 = abs(lam - 1)
    base = np.maximum(x * f32(lam) / f32(lam_1) + f32(1), f32(eps))
    return ((_pow_like_aten(base, 1 / lam) - f32(1)) * f32(lam_1)).astype(f32)


@dataclass
class Spacing:
    """the ``spacing_to_euclidean_fn`` closure of ray_samplers.py:117-118."""

    s_near: np.ndarray  # [R,1]
    s_far: np.ndarray
    lam: float
    scaling: float

    def __call__(self, x):
        x = np.asarray(x, f32)
        return (inv_power_fn(x * self.s_far + (f32(1) - x) * self.s_near, self.lam) / f32(self.scaling)).astype(f32)


def power_sampler(nears, fars, num_samples: int, lam: float = -1.0, scaling: float = 0.1, t_rand=None):
    """-> (spacing bins [R,S+1], euclidean bins [R,S+1], Spacing).  ``t_rand`` [R,S+1] injects the
    training-mode stratified jitter (ray_samplers.py:104-112); None = eval mode."""
    nears = np.asarray(nears, f32).reshape(-1, 1)
    fars = np.asarray(fars, f32).reshape(-1, 1)
    bins = np.linspace(0.0, 1.0, num_samples + 1, dtype=f32)[None, :]
    if t_rand is not None:
        centers = (bins[..., 1:] + bins[..., :-1]) / f32(2)
        upper = np.concatenate([centers, bins[..., -1:]], -1)
        lower = np.concatenate([bins[..., :1], centers], -1)
        bins = lower + (upper - lower) * np.asarray(t_rand, f32)
    sp = Spacing(power_fn(nears * f32(scaling), lam), power_fn(fars * f32(scaling), lam), lam, scaling)
    bins = np.broadcast_to(bins, (nears.shape[0], num_samples + 1)).astype(f32)
    return bins, sp(bins), sp


# --------------------------------------------------------------------------------------
# S4  PDFSampler.generate_ray_samples (ray_samplers.py:280-376), include_original=False
# --------------------------------------------------------------------------------------
def pdf_sample(weights, spacing_bins, num_samples: int, spacing: Spacing, histogram_padding: float = 0.01,
               eps: float = 1e-5, rand=None):
    """weights [R,Sp], spacing_bins [R,Sp+1] -> (new spacing bins [R,num_samples+1], euclidean bins).
    ``rand`` [R,1] (single_jitter) or [R,num_samples+1] = training-mode jitter in [0,1); None = eval."""
    w = np.asarray(weights, f32) + f32(histogram_padding)
    wsum = w.sum(-1, keepdims=True, dtype=f32)
    padding = np.maximum(f32(eps) - wsum, f32(0))
    w = w + padding / f32(w.shape[-1])
    wsum = wsum + padding
    pdf = w / wsum
    cdf = np.minimum(f32(1), np.cumsum(pdf.astype(np.float64), -1).astype(f32))
    cdf = np.concatenate([np.zeros_like(cdf[..., :1]), cdf], -1)
    nb = num_samples + 1
    u = np.linspace(0.0, 1.0 - (1.0 / nb), nb, dtype=f32)
    if rand is not None:
        u = u[None, :] + np.asarray(rand, f32) / f32(nb)
    else:
        u = u + f32(1.0 / (2 * nb))
    u = np.broadcast_to(u, (cdf.shape[0], nb)).astype(f32)
    existing = np.asarray(spacing_bins, f32)
    inds = np.stack([np.searchsorted(cdf[r], u[r], side="right") for r in range(cdf.shape[0])])
    hi = existing.shape[-1] - 1
    below = np.clip(inds - 1, 0, hi)
    above = np.clip(inds, 0, hi)
    cdf0 = np.take_along_axis(cdf, below, -1)
    b0 = np.take_along_axis(existing, below, -1)
    cdf1 = np.take_along_axis(cdf, above, -1)
    b1 = np.take_along_axis(existing, above, -1)
    with np.errstate(divide="ignore", invalid="ignore"):
        t = (u - cdf0) / (cdf1 - cdf0)
    t = np.clip(np.nan_to_num(t, nan=0.0), 0, 1).astype(f32)
    bins = (b0 + t * (b1 - b0)).astype(f32)
    return bins, spacing(bins)


# --------------------------------------------------------------------------------------
# S5 + M1  ProposalNetworkSampler (ray_samplers.py:623-666) driven as NeuRADModel._get_ray_samples
#          (models/neurad.py:443-459), incl. the late-binding quirk of models/neurad.py:248
# --------------------------------------------------------------------------------------
@dataclass
class SamplerOutput:
    starts: np.ndarray  # [R,S] final euclidean starts (sky-stretched)
    ends: np.ndarray
    spacing_starts: np.ndarray
    spacing_ends: np.ndarray
    prop_weights: List[np.ndarray] = field(default_factory=list)
    prop_starts: List[np.ndarray] = field(default_factory=list)
    prop_ends: List[np.ndarray] = field(default_factory=list)
    prop_spacing: List[np.ndarray] = field(default_factory=list)


def proposal_sampler(props: Sequence[ProposalParams], origins, directions, pixel_area, nears, fars,
                     num_proposal_samples=(128, 64), num_nerf_samples=32, lam=-1.0, scaling=0.1,
                     sky_distance=20000.0, late_binding_quirk=True, stretch_sky=True, rands=None) -> SamplerOutput:
    """eval-mode (deterministic) sampler chain unless ``rands`` = [t_rand0, rand1, rand2] is given.

    ``late_binding_quirk``: the reference's ``density_fns`` list comprehension closes over the loop
    variable, so BOTH rounds evaluate ``proposal_fields[-1]`` (models/neurad.py:248).
    """
    fars = np.minimum(np.asarray(fars, f32), f32(sky_distance))  # neurad.py:445-446
    n = len(num_proposal_samples)
    rands = rands if rands is not None else [None] * (n + 1)
    bins, eu, sp = power_sampler(nears, fars, num_proposal_samples[0], lam, scaling, rands[0])
    out_w, out_s, out_e, out_sp = [], [], [], []
    for i in range(n):
        p = props[-1] if late_binding_quirk else props[i]
        dens = proposal_density(p, origins, directions, pixel_area, eu[:, :-1], eu[:, 1:])
        w = weights_from_density(eu[:, 1:] - eu[:, :-1], dens)
        out_w.append(w), out_s.append(eu[:, :-1]), out_e.append(eu[:, 1:]), out_sp.append(bins)
        ns = num_proposal_samples[i + 1] if i + 1 < n else num_nerf_samples
        bins, eu = pdf_sample(w, bins, ns, sp, rand=rands[i + 1])  # anneal = 1 -> pow(w,1) no-op
    starts, ends = eu[:, :-1].copy(), eu[:, 1:].copy()
    sps, spe = bins[:, :-1].copy(), bins[:, 1:].copy()
    if stretch_sky:  # neurad.py:451-455
        ends[:, -1] += f32(sky_distance) - ends[:, -1]
        spe[:, -1] = f32(1 - 1e-7)
    return SamplerOutput(starts, ends, sps, spe, out_w, out_s, out_e, out_sp)


# --------------------------------------------------------------------------------------
# get_nff_outputs end to end (models/neurad.py:368-398), no appearance embedding
# --------------------------------------------------------------------------------------
def render_rays(p: FieldParams, origins, directions, pixel_area, starts, ends):
    f = field_fwd(p, origins, directions, pixel_area, starts, ends)
    if p.use_sdf:
        w, _ = render_weight_from_alpha(f["alpha"])
    else:
        w, _, _ = render_weight_from_density(starts, ends, f["density"])
    feat, depth, acc = composite(w, f["feature"], starts, ends)
    return {"features": feat, "depth": depth, "accumulation": acc, "weights": w, **f}


# --------------------------------------------------------------------------------------
# H5  dynamic actors (field_components/neurad_encoding.py:150-295; model_components/dynamic_actors.py:106-108,
#     251-268; utils/poses.py:42-55,90-150; cameras/camera_utils.py:422-443; cameras/lidars.py:550-564)
# --------------------------------------------------------------------------------------
def _normalize(v, eps=1e-12):  # torch.nn.functional.normalize
    n = np.sqrt((v * v).sum(-1, keepdims=True, dtype=f32)).astype(f32)
    return (v / np.maximum(n, f32(eps))).astype(f32)


def rotation_6d_to_matrix(d6):
    """camera_utils.py:422-443 (rows b1, b2, b3)."""
    a1, a2 = d6[..., :3], d6[..., 3:]
    b1 = _normalize(a1)
    b2 = _normalize(a2 - (b1 * a2).sum(-1, keepdims=True) * b1)
    b3 = np.cross(b1, b2)
    return np.stack([b1, b2, b3], axis=-2).astype(f32)


@dataclass
class ActorParams:
    """state of DynamicActors (model_components/dynamic_actors.py:109-170) + the per-actor grids of the torch path."""

    timestamps: np.ndarray        # [Tn]
    positions: np.ndarray         # [Tn, A, 3]   actor_positions
    rotations_6d: np.ndarray      # [Tn, A, 6]   actor_rotations_6d
    present: np.ndarray           # [Tn, A] bool actor_present_at_time
    sizes: np.ndarray             # [A, 3]       actor_sizes (wlh)
    padding: np.ndarray           # [3]          actor_padding
    grids: List[GridParams]       # one 3-D grid per actor (neurad_encoding.py:110-131)
    actor_scale: float = 10.0

    @property
    def bounds(self):
        return (self.sizes / f32(2) + self.padding).astype(f32)  # dynamic_actors.py:106-107


def edit_boxes2world(b2w, edit):
    """DynamicActors.edit_boxes2world, the flatten=False branch (model_components/dynamic_actors.py:181-249), applied by
    get_boxes2world outside training (:261-265).  b2w [R,A,4,4] (only the 3x4 part is read / written), edit = the
    ``actor_editing`` dict (:53-59).  In place, like the reference."""
    if abs(edit["longitudinal"]) == 0.0 and abs(edit["lateral"]) == 0.0 and abs(edit["rotation"]) == 0.0:
        return b2w  # (:182-187: a height-only edit changes nothing)
    A = b2w.shape[1]
    idx = np.arange(A) if edit["index"] == -1.0 else np.array([int(min(edit["index"], A - 1))])  # (:189-193)
    if abs(edit["longitudinal"]) > 0.0 or abs(edit["lateral"]) > 0.0 or abs(edit.get("height", 0.0)) > 0.0:
        v = np.array([edit["lateral"], edit["longitudinal"], edit.get("height", 0.0), 1.0], f32)
        for i in idx:
            b2w[:, i, :3, 3] = (b2w[:, i, :3, :] @ v).astype(f32)  # (:216-227)
    if abs(edit["rotation"]) > 0.0:
        c, s = math.cos(edit["rotation"]), math.sin(edit["rotation"])
        yaw = np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]], f32)
        for i in idx:
            b2w[:, i, :3, :3] = (yaw @ b2w[:, i, :3, :3]).astype(f32)  # (:238-249)
    return b2w


def actor_boxes2world(a: ActorParams, query_times, edit=None):
    """DynamicActors.get_boxes2world(flatten=False) -> (boxes2world [R,A,4,4], valid [R,A]);
    interpolate_trajectories_6d (utils/poses.py:90-150).  edit: the eval-time ``actor_editing`` dict or None."""
    poses = np.concatenate([a.rotations_6d, a.positions], -1).astype(f32)  # [Tn,A,9]
    a1 = _normalize(poses[..., :3])
    a2 = poses[..., 3:6]
    a2 = _normalize(a2 - (a1 * a2).sum(-1, keepdims=True) * a1)
    poses = np.concatenate([a1, a2, poses[..., 6:9]], -1)
    q = np.asarray(query_times, f32).reshape(-1)
    right = np.searchsorted(a.timestamps, q, side="left")
    left = np.clip(right - 1, 0, None)
    right = np.clip(right, None, len(a.timestamps) - 1)
    rt, lt = a.timestamps[right].astype(f32), a.timestamps[left].astype(f32)
    frac = np.clip((q - lt) / (rt - lt + f32(1e-6)), 0, 1).astype(f32)
    valid = a.present[left] | a.present[right]
    pl, pr = poses[left], poses[right]
    interp = (pl + (pr - pl) * frac[:, None, None]).astype(f32)  # [R,A,9]
    rot = rotation_6d_to_matrix(interp[..., :6])
    b2w = np.zeros(interp.shape[:2] + (4, 4), f32)
    b2w[..., :3, :3] = rot
    b2w[..., :3, 3] = interp[..., 6:]
    b2w[..., 3, 3] = 1
    if edit is not None:
        b2w = edit_boxes2world(b2w, edit)
    return b2w, valid


def pose_inverse(b2w):
    """utils/poses.py:42-55 -> [...,3,4]"""
    R = b2w[..., :3, :3]
    t = b2w[..., :3, 3:]
    Ri = np.swapaxes(R, -1, -2)
    return np.concatenate([Ri, -(Ri @ t)], -1).astype(f32)


def actor_hits(a: ActorParams, mean_pos, b2w, valid, w2b):
    """_get_actor_indices (neurad_encoding.py:225-263) -> (ray_idx, sample_idx, actor_idx), duplicates kept, in the
    reference's order (pair order, then sample)."""
    eps = f32(1e-7)
    bounds = a.bounds
    radii = np.sqrt((bounds * bounds).sum(-1)).astype(f32)
    p0 = mean_pos[:, 0, :]
    ld = mean_pos[:, -1, :] - p0
    ld = ld / (np.linalg.norm(ld, axis=-1, keepdims=True).astype(f32) + eps)
    vec = b2w[..., :3, 3] - p0[:, None, :]
    dist = np.linalg.norm(np.cross(vec, ld[:, None, :]), axis=-1).astype(f32)
    close = (dist < radii[None, :]) & valid
    ray_idx, actor_idx = np.nonzero(close)
    sp = mean_pos[ray_idx]                                     # [P,S,3]
    ap = b2w[ray_idx, actor_idx, :3, 3][:, None, :]
    d2 = np.linalg.norm(sp - ap, axis=-1).astype(f32)
    within = d2 < radii[actor_idx][:, None]
    pi, si = np.nonzero(within)
    r, s, k = ray_idx[pi], si, actor_idx[pi]
    sel = mean_pos[r, s]
    tr = w2b[r, k]
    pib = (tr[:, :3, :3] @ sel[:, :, None])[:, :, 0] + tr[:, :3, 3]
    inside = np.all(np.abs(pib) < bounds[k], axis=-1)
    return r[inside], s[inside], k[inside]


def encode_with_actors(grid: GridParams, static_scale, a: ActorParams, origins, directions, pixel_area, starts, ends,
                       times, ray_flip=None, edit=None):
    """NeuRADHashEncoding.forward (neurad_encoding.py:150-187), eval mode (no flip), per-actor 3-D grids.
    -> features [R*S, L*F], directions [R,S,3] (box frame + renormalised where a sample is inside an actor)."""
    R, S = np.asarray(starts).shape
    mean, std = fast_isotropic_gaussian(origins, directions, pixel_area, starts, ends)
    feats = encode_static(grid, static_scale, origins, directions, pixel_area, starts, ends).reshape(R, S, -1).copy()
    dirs = np.broadcast_to(np.asarray(directions, f32)[:, None, :], (R, S, 3)).copy()
    if len(a.grids) == 0:
        return feats.reshape(R * S, -1), dirs
    b2w, valid = actor_boxes2world(a, times, edit)
    w2b = pose_inverse(b2w)
    r, s, k = actor_hits(a, mean, b2w, valid, w2b)
    if r.shape[0] == 0:
        return feats.reshape(R * S, -1), dirs
    tr = w2b[r, k]
    pos = (tr[:, :3, :3] @ mean[r, s][:, :, None])[:, :, 0] + tr[:, :3, 3]
    dd = (tr[:, :3, :3] @ dirs[r, s][:, :, None])[:, :, 0]
    dd = dd / (np.linalg.norm(dd, axis=-1, keepdims=True).astype(f32) + f32(1e-7))
    if ray_flip is not None:  # training-mode per-ray x flip, injected (+-1 per ray), neurad_encoding.py:212-219
        fl = np.asarray(ray_flip, f32)[r]
        pos[:, 0] *= fl
        dd[:, 0] *= fl
    dirs[r, s] = dd  # last write wins (CPU index_put order), as for the features below
    cpos, cstd = contract_gaussian(pos.astype(f32), std[r, s], a.actor_scale)
    out_dim = feats.shape[-1]
    for m in range(r.shape[0]):  # reference order: later triples overwrite earlier ones (neurad_encoding.py:184-185)
        g = a.grids[k[m]]
        f = hashgrid_fwd(cpos[m:m + 1], g.table, g.scalings, g.table_size)
        f = rescale_grid_features(f, cstd[m:m + 1], g.scalings, g.n_feat)[0]
        feats[r[m], s[m]] = np.pad(f, (0, out_dim - f.shape[0]))
    return feats.reshape(R * S, -1), dirs


def field_fwd_actors(p: FieldParams, a: ActorParams, origins, directions, pixel_area, starts, ends, times, edit=None):
    """NeuRADField.forward with dynamic actors (fields/neurad_field.py:128-152)."""
    R, S = np.asarray(starts).shape
    enc, dirs = encode_with_actors(p.grid, p.static_scale, a, origins, directions, pixel_area, starts, ends, times,
                                   edit=edit)
    geo = mlp_fwd(enc, p.geo_w, p.geo_b)
    geo_out, geo_emb = geo[:, :1], geo[:, 1:]
    sh = sh_deg4(((dirs + f32(1)) / f32(2)).reshape(-1, 3))
    feat = geo_emb + mlp_fwd(np.concatenate([geo_emb, sh], -1), p.feat_w, p.feat_b)
    out = {"feature": feat.reshape(R, S, -1).astype(f32), "directions": dirs, "enc": enc}
    if p.use_sdf:
        beta = f32(abs(p.beta) + p.beta_min)
        out["sdf"] = geo_out.reshape(R, S)
        out["alpha"] = sigmoid(-geo_out.reshape(R, S) * beta)
    else:
        out["density"] = np.exp(geo_out.reshape(R, S)).astype(f32)
    return out


# ------------------------------------------------------------------------------------------------------------------
# losses on the sampler outputs (model_components/losses.py)
def blur_stepfun(x: np.ndarray, y: np.ndarray, r: float):
    """losses.py:645-653 for one ray: x [n+1] edges, y [n] step heights -> (xr [2n+2], yr [2n+2])."""
    xr_all = np.concatenate([x - f32(r), x + f32(r)]).astype(f32)
    idx = np.argsort(xr_all, kind="stable")
    xr = xr_all[idx]
    y1 = ((np.concatenate([y, [f32(0)]]) - np.concatenate([[f32(0)], y])) / f32(2 * r)).astype(f32)
    y2 = np.concatenate([y1, -y1])[idx[:-1]]
    # torch's CPU cumsum accumulates float32 inputs in double (at::acc_type) and rounds every prefix to float32
    inner = np.cumsum(y2, dtype=np.float64).astype(f32)
    yr = np.maximum(np.cumsum(((xr[1:] - xr[:-1]) * inner).astype(f32), dtype=np.float64).astype(f32), f32(0))
    return xr, np.concatenate([[f32(0)], yr]).astype(f32)


def sorted_interp_quad(x, xp, fpdf, fcdf):
    """losses.py:656-669 for one ray."""
    right = np.searchsorted(xp, x, side="left")
    left = np.maximum(right - 1, 0)
    right = np.minimum(right, xp.shape[-1] - 1)
    xp0, xp1, f0, f1, c0 = xp[left], xp[right], fpdf[left], fpdf[right], fcdf[left]
    with np.errstate(divide="ignore", invalid="ignore"):
        off = (x - xp0) / (xp1 - xp0)
    off = np.clip(np.nan_to_num(off, nan=0.0), 0, 1).astype(f32)
    return (c0 + (x - xp0) * (f0 + f1 * off + f0 * (1 - off)) * f32(0.5)).astype(f32)


def interlevel_loss_level(c, w, cp, wp, pulse_width):
    """One proposal level of zipnerf_interlevel_loss (losses.py:672-705).  c [R,Sf+1], w [R,Sf] fine spacing edges /
    weights, cp [R,Sp+1], wp [R,Sp] proposal ones.  -> per-ray loss [R], w_s [R,Sp], d loss_ray / d wp [R,Sp]."""
    c, w, cp, wp = (np.asarray(a, f32) for a in (c, w, cp, wp))
    R = c.shape[0]
    loss, ws_all, g_all = np.zeros(R, f32), np.zeros_like(wp), np.zeros_like(wp)
    for i in range(R):
        wi = w[i].copy()
        wi[-1] = wi[-1] + (f32(1) - wi.sum(dtype=f32))
        wn = (wi / (c[i, 1:] - c[i, :-1])).astype(f32)
        c_, w_ = blur_stepfun(c[i], wn, pulse_width)
        area = (f32(0.5) * (w_[1:] + w_[:-1]) * (c_[1:] - c_[:-1])).astype(f32)
        cdf = np.concatenate([[f32(0)], np.cumsum(area, dtype=np.float64).astype(f32)])
        c_ = np.concatenate([[f32(0)], c_, [f32(1)]]).astype(f32)
        w_ = np.concatenate([[f32(0)], w_, [f32(0)]]).astype(f32)
        cdf = np.concatenate([[f32(0)], cdf, [f32(1)]]).astype(f32)
        ci = sorted_interp_quad(cp[i], c_, w_, cdf)
        ws = np.diff(ci).astype(f32)
        d = np.maximum(ws - wp[i], f32(0))
        den = wp[i] + f32(1e-5)
        loss[i] = (d * d / den).sum(dtype=f32)
        ws_all[i] = ws
        g_all[i] = -(f32(2) * d / den + d * d / (den * den))
    return loss, ws_all, g_all


def distortion_loss_rays(c, w):
    """lossfun_distortion (losses.py:137-148) per ray: c [R,S+1], w [R,S] -> loss [R], d loss / d w [R,S]."""
    c, w = np.asarray(c, f32), np.asarray(w, f32)
    ut = ((c[:, 1:] + c[:, :-1]) / f32(2)).astype(f32)
    dut = np.abs(ut[:, :, None] - ut[:, None, :]).astype(f32)
    inner = (w[:, None, :] * dut).sum(-1, dtype=f32)
    delta = (c[:, 1:] - c[:, :-1]).astype(f32)
    loss = (w * inner).sum(-1, dtype=f32) + (w * w * delta).sum(-1, dtype=f32) / f32(3)
    grad = f32(2) * inner + f32(2) / f32(3) * w * delta
    return loss.astype(f32), grad.astype(f32)


# ------------------------------------------------------------------------------------------------------------------
# ScaledPatchSampler: patch centres -> ray indices + rgb patches (nerfstudio/data/pixel_samplers.py:618-742)
# ------------------------------------------------------------------------------------------------------------------
def patch_centers_from_uniforms(uniforms, n_images: int, height: int, width: int, rgb_size: int):
    """PixelSampler.sample_method on the cropped extent (pixel_samplers.py:100-103) + the crop shift of
    ScaledPatchSampler.sample_method (:722-726): (u * float(dims)).long(), then + rgb_size // 2 on (y, x)."""
    dims = np.array([n_images, height - rgb_size + 1, width - rgb_size + 1], dtype=np.float32)
    c = (np.asarray(uniforms, dtype=np.float32) * dims).astype(np.int64)  # fp32 product, truncation
    c[:, 1:] += rgb_size // 2
    return c


def patches_from_centers(images, centers, patch_size: int, patch_scale: int, image_idx=None):
    """ScaledPatchSampler._patches_from_centers (pixel_samplers.py:696-714) + the global image index (:652).
    images [N,H,W,C]; centers [P,3] int64 (image, y, x).  Returns (ray_indices [P*patch_size^2, 3] int64,
    coords [P*patch_size^2, 2] fp32 = RayGenerator's image_coords[y, x] (ray_generators.py:49), patches [P,K,K,C])."""
    K = patch_size * patch_scale
    off = np.arange(-(K // 2), K // 2 + K % 2)
    yy, xx = np.meshgrid(off, off, indexing="ij")
    centers = np.asarray(centers, dtype=np.int64)
    img = np.broadcast_to(centers[:, None, None, 0], (centers.shape[0], K, K))
    ys = centers[:, None, None, 1] + yy[None]
    xs = centers[:, None, None, 2] + xx[None]
    sl = slice(patch_scale // 2, None, patch_scale)
    rays = np.stack([img[:, sl, sl], ys[:, sl, sl], xs[:, sl, sl]], -1).reshape(-1, 3)
    patches = None if images is None else np.asarray(images)[img, ys, xs]
    coords = np.stack([rays[:, 1].astype(np.float32) + 0.5, rays[:, 2].astype(np.float32) + 0.5], -1)
    if image_idx is not None:
        rays = rays.copy()
        rays[:, 0] = np.asarray(image_idx, dtype=np.int64)[rays[:, 0]]
    return rays, coords, patches


def lidar_point_sample(lidar, points_per_lidar, num_rays: int, shuffle, draws, lidar_idx=None):
    """LidarPointSampler.collate_image_dataset_batch (pixel_samplers.py:538-583) given torch's draws: shuffle =
    randperm(n) (:540), draws = rand((n, ceil(num_rays / n)), float64) (:552).  Returns (indices [num_rays,2], points)."""
    npl = np.asarray(points_per_lidar, dtype=np.int64)
    n = npl.shape[0]
    first = np.zeros(n, dtype=np.int64)
    first[1:] = np.cumsum(npl)[:-1]
    point = np.floor(np.asarray(draws, dtype=np.float64) * npl[:, None].astype(np.float64)).astype(np.int64)
    scan = np.repeat(np.arange(n)[:, None], point.shape[1], 1)
    shuffle = np.asarray(shuffle)
    scan, point, first = scan[shuffle], point[shuffle], first[shuffle]
    flat = (point + first[:, None]).reshape(-1)[:num_rays]
    idx = np.stack([scan.reshape(-1), point.reshape(-1)], -1)[:num_rays]
    if lidar_idx is not None:
        idx[:, 0] = np.asarray(lidar_idx, dtype=np.int64)[idx[:, 0]]
    return idx, np.asarray(lidar)[flat]
