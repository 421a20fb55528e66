"""TEST INFRASTRUCTURE / CPU BASELINE ONLY -- never imported by the product package.

The reference's field-eval path as torch ops on CPU tensors: the op sequence of HashEncoding.pytorch_fwd
(field_components/encodings.py:425-466: scale, floor / ceil corners, xor-prime hash, eight gathers, trilinear blend),
Frustums.get_fast_isotropic_gaussian (cameras/rays.py:109-124), ScaledSceneContraction (spatial_distortions.py:103-141), the
feature rescaling of NeuRADHashEncoding (neurad_encoding.py:297-304), MLP.pytorch_fwd (mlp.py:159-178: F.linear + ReLU),
SHEncoding's torch branch (utils/math.py:31-94), SigmoidDensity (model_components/utils.py:21-41) and the dense compositing of
get_nff_outputs (models/neurad.py:377-395).  It exists because the GPU box carries no reference tree: `bench.py` times THIS on
the box's host cores as the `reference_torch_cpu` leg when the reference itself cannot be imported (kind "port", torch ops,
all host threads), next to the figure measured with the reference in the build container.  Pinned to the numpy oracle -- and
through it to the reference's golden vectors -- by tests/test_oracle_torch_port.py."""
from __future__ import annotations

import torch
import torch.nn.functional as F

PRIME_Y, PRIME_Z = 2654435761, 805459861  # encodings.py:419


def hashgrid_fwd(x: torch.Tensor, table: torch.Tensor, scalings: torch.Tensor, table_size: int) -> torch.Tensor:
    """x [N,3] in [0,1], table [L*T, F] -> [N, L*F]"""
    L = scalings.shape[0]
    scaled = x[:, None, :] * scalings.view(L, 1)
    c, f = torch.ceil(scaled).to(torch.int32), torch.floor(scaled).to(torch.int32)
    o = scaled - f.to(scaled.dtype)
    off = (torch.arange(L, dtype=torch.int64) * table_size)[None, :]

    def h(sx, sy, sz):
        a, b, cc = sx[..., 0].long(), sy[..., 1].long(), sz[..., 2].long()
        return torch.remainder(a ^ (b * PRIME_Y) ^ (cc * PRIME_Z), table_size) + off

    combos = [(c, c, c), (c, f, c), (f, f, c), (f, c, c), (c, c, f), (c, f, f), (f, f, f), (f, c, f)]
    fc = [table[h(*cb)] for cb in combos]
    ox, oy, oz = o[..., 0:1], o[..., 1:2], o[..., 2:3]
    f03, f12 = fc[0] * ox + fc[3] * (1 - ox), fc[1] * ox + fc[2] * (1 - ox)
    f56, f47 = fc[5] * ox + fc[6] * (1 - ox), fc[4] * ox + fc[7] * (1 - ox)
    enc = (f03 * oy + f12 * (1 - oy)) * oz + (f47 * oy + f56 * (1 - oy)) * (1 - oz)
    return enc.reshape(x.shape[0], -1)


def sh_deg4(d: torch.Tensor) -> torch.Tensor:
    x, y, z = d[..., 0], d[..., 1], d[..., 2]
    xx, yy, zz = x * x, y * y, z * z
    return torch.stack([
        torch.full_like(x, 0.28209479177387814), 0.4886025119029199 * y, 0.4886025119029199 * z, 0.4886025119029199 * x,
        1.0925484305920792 * x * y, 1.0925484305920792 * y * z, 0.9461746957575601 * zz - 0.31539156525251999,
        1.0925484305920792 * x * z, 0.5462742152960396 * (xx - yy), 0.5900435899266435 * y * (3 * xx - yy),
        2.890611442640554 * x * y * z, 0.4570457994644658 * y * (5 * zz - 1), 0.3731763325901154 * z * (5 * zz - 3),
        0.4570457994644658 * x * (5 * zz - 1), 1.445305721320277 * z * (xx - yy), 0.5900435899266435 * x * (xx - 3 * yy)], -1)


def mlp(x, weights, biases):
    n = len(weights)
    for i, (w, b) in enumerate(zip(weights, biases)):
        x = F.linear(x, w, b)
        if i < n - 1:
            x = torch.relu(x)
    return x


@torch.no_grad()
def render_rays(p, origins, directions, pixel_area, starts, ends):
    """p: oracle.neurad_oracle.FieldParams (numpy) or the same fields as tensors -> features [R,C], depth, accumulation"""
    t = lambda a: a if isinstance(a, torch.Tensor) else torch.from_numpy(a)  # noqa: E731
    o, d, area, s, e = (t(a).float() for a in (origins, directions, pixel_area, starts, ends))
    R, S = s.shape
    g = p.grid
    table, scal = t(g.table).float(), t(g.scalings).float()
    # H2 Gaussians
    dist = (e - s) / 2
    tm = s + dist
    mean = o[:, None, :] + d[:, None, :] * tm[..., None]
    std = torch.pow(area.view(-1, 1) * tm.pow(2) * dist, 1 / 3)
    # H3 contraction
    m, sd = mean / p.static_scale, std / p.static_scale
    mag = m.abs().amax(-1, keepdim=True)
    inside = mag < 1
    cm = mag.clamp(min=1)
    m2 = torch.where(inside, m, (2 - 1 / cm) * (m / cm))
    sc = (torch.pow(2 * cm - 1, 1 / 3) / cm) ** 2
    sd2 = torch.where(inside[..., 0], sd, sd * sc[..., 0])
    pos, cstd = (m2 + 2) / 4, sd2 / 4
    # H1 + H4
    feat = hashgrid_fwd(pos.reshape(-1, 3), table, scal, g.table_size)
    L = scal.shape[0]
    w = 1 / torch.clamp(scal[None, :] * 2 * cstd.reshape(-1, 1), min=1)
    enc = (feat.view(-1, L, g.n_feat) * w[..., None]).reshape(-1, L * g.n_feat)
    # F1-F4
    geo = mlp(enc, [t(a).float() for a in p.geo_w], [t(a).float() for a in p.geo_b])
    geo_out, geo_emb = geo[:, :1], geo[:, 1:]
    sh = sh_deg4(((d + 1) / 2)[:, None, :].expand(R, S, 3).reshape(-1, 3))
    feature = geo_emb + mlp(torch.cat([geo_emb, sh], -1), [t(a).float() for a in p.feat_w], [t(a).float() for a in p.feat_b])
    feature = feature.view(R, S, -1)
    if p.use_sdf:
        alpha = torch.sigmoid(-geo_out.view(R, S) * (abs(p.beta) + p.beta_min))
        trans = torch.cat([torch.ones_like(alpha[:, :1]), torch.cumprod(1 - alpha[:, :-1], -1)], -1)
        wts = trans * alpha
    else:
        sig = torch.exp(geo_out.view(R, S)) * (e - s)
        trans = torch.exp(-torch.cat([torch.zeros_like(sig[:, :1]), torch.cumsum(sig[:, :-1], -1)], -1))
        wts = trans * (1 - torch.exp(-sig))
    acc = wts.sum(-1, keepdim=True)
    w2 = torch.cat([wts[:, :-1], wts[:, -1:] + 1 - acc], -1)
    feats = (w2[..., None] * feature).sum(-2)
    steps = (s + e) / 2
    depth = (w2[:, :-1] * steps[:, :-1]).sum(-1, keepdim=True)
    return {"features": feats, "depth": depth, "accumulation": acc}
