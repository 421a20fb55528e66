"""Generate tests/golden/*.npz by RUNNING THE REFERENCE ITSELF (its implementation="torch" path).

Run in the build container only (needs /root/reference):  python oracle/make_golden.py
The GPU box never runs this; it only reads the committed fixtures.
Inputs/weights come from tests/synth.py (deterministic integer hash), so fixtures hold only
small inputs and the reference's outputs.
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np
import torch

import ref_import

ref_import.install()
import synth  # noqa: E402
from nerfstudio.cameras.rays import RayBundle  # noqa: E402
from nerfstudio.field_components.encodings import HashEncoding, SHEncoding  # noqa: E402
from nerfstudio.field_components.field_heads import FieldHeadNames  # noqa: E402
from nerfstudio.field_components.mlp import MLP  # noqa: E402
from nerfstudio.field_components.spatial_distortions import ScaledSceneContraction  # noqa: E402
from nerfstudio.fields.neurad_field import (  # noqa: E402
    NeuRADField, NeuRADFieldConfig, NeuRADProposalField, NeuRADProposalFieldConfig)
from nerfstudio.field_components.neurad_encoding import (  # noqa: E402
    ActorSettings, NeuRADHashEncodingConfig, StaticSettings)
from nerfstudio.model_components.dynamic_actors import DynamicActors, DynamicActorsConfig  # noqa: E402
from nerfstudio.model_components.ray_samplers import PDFSampler, PowerSampler, ProposalNetworkSampler  # noqa: E402
from nerfstudio.utils.math import GaussiansStd  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)
T = torch.from_numpy


def save(name, **kw):
    kw = {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in kw.items()}
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **kw)
    print(name, {k: v.shape for k, v in kw.items()})


def set_linear(layer, seed, bias=True):
    w, b = synth.linear(layer.out_features, layer.in_features, seed, bias)
    layer.weight.data = T(w)
    if bias:
        layer.bias.data = T(b)


def golden_hashgrid():
    for tag, (L, mn, mx, lg, F) in {
        "c2small": (16, 16, 1024, 12, 2), "neurad": (8, 32, 8192, 12, 4), "prop": (6, 128, 4096, 11, 1),
        "tiny": (1, 32, 32, 10, 4), "actor": (4, 64, 1024, 10, 4),
    }.items():
        enc = HashEncoding(num_levels=L, min_res=mn, max_res=mx, log2_hashmap_size=lg, features_per_level=F,
                           implementation="torch")
        enc.hash_table.data = T(synth.hash_table(L * 2**lg, F, seed=11))
        x = synth.uniform((300, 3), 0.0, 1.0, seed=3)
        x[:4] = np.array([[0, 0, 0], [1, 1, 1], [0.5, 0.25, 0.125], [1, 0, 0.5]], np.float32)  # lattice hits
        xt = T(x)
        y = enc(xt)
        # exact corner index of the all-floor corner (hashed_6) and all-ceil (hashed_0) for bit-exact checks
        scaled = xt[:, None, :] * enc.scalings.view(-1, 1)
        h6 = enc.hash_fn(torch.floor(scaled).type(torch.int32))
        h0 = enc.hash_fn(torch.ceil(scaled).type(torch.int32))
        g = T(synth.normal(tuple(y.shape), seed=5))
        (y * g).sum().backward()
        save(f"hashgrid_{tag}", cfg=np.array([L, mn, mx, lg, F]), x=x, y=y, scalings=enc.scalings, h_floor=h6,
             h_ceil=h0, grad_out=g, grad_table_nz_idx=enc.hash_table.grad.abs().sum(-1).nonzero()[:, 0],
             grad_table_nz=enc.hash_table.grad[enc.hash_table.grad.abs().sum(-1) > 0])
    # scalings of the full-size configs (BASELINE configs)
    sc = {}
    for tag, (L, mn, mx) in {"c2": (16, 16, 1024), "neurad": (8, 32, 8192), "prop": (6, 128, 4096),
                             "actor": (4, 64, 1024), "neurader": (8, 64, 16384)}.items():
        sc[tag] = HashEncoding(num_levels=L, min_res=mn, max_res=mx, log2_hashmap_size=4,
                               implementation="torch").scalings
    save("scalings", **sc)


def golden_sh_mlp_misc():
    d = synth.normal((128, 3), seed=21)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    sh = SHEncoding(levels=4, implementation="torch")
    save("sh", d01=(d + 1) / 2, y=sh(T((d + 1) / 2)))
    for tag, (i, n, w, o) in {"geo64": (32, 2, 64, 33), "feat64": (48, 3, 64, 32), "geo32": (32, 2, 32, 33),
                              "lidar": (48, 3, 32, 2)}.items():
        mlp = MLP(in_dim=i, num_layers=n, layer_width=w, out_dim=o, implementation="torch")
        for k, layer in enumerate(mlp.layers):
            set_linear(layer, 100 + 10 * k)
        x = T(synth.normal((200, i), seed=31) * 0.05)
        x.requires_grad_(True)
        y = mlp(x)
        g = T(synth.normal(tuple(y.shape), seed=33))
        (y * g).sum().backward()
        save(f"mlp_{tag}", cfg=np.array([i, n, w, o]), x=x, y=y, grad_out=g, dx=x.grad,
             **{f"dw{k}": l.weight.grad for k, l in enumerate(mlp.layers)},
             **{f"db{k}": l.bias.grad for k, l in enumerate(mlp.layers)})
    # contraction of a GaussiansStd
    mean = synth.normal((500, 3), seed=41) * 120.0
    mean[:50] *= 0.01
    std = synth.uniform((500, 1), 1e-4, 3.0, seed=42)
    c = ScaledSceneContraction(order=float("inf"), scale=80.0)(GaussiansStd(mean=T(mean), std=T(std)))
    save("contraction", mean=mean, std=std, scale=np.float32(80.0), cmean=c.mean, cstd=c.std)


def make_bundle(R, seed, fars=None):
    o, d, area, t = synth.rays(R, seed)
    rb = RayBundle(origins=T(o), directions=T(d), pixel_area=T(area)[:, None], times=T(t)[:, None],
                   nears=torch.zeros(R, 1), fars=torch.full((R, 1), 20000.0) if fars is None else T(fars)[:, None])
    return rb, (o, d, area, t)


def no_actors():
    return DynamicActors(DynamicActorsConfig(), trajectories=[])


def golden_field():
    for tag, cfg in {
        "sdf": dict(use_sdf=True), "density": dict(use_sdf=False),
    }.items():
        grid = NeuRADHashEncodingConfig(static=StaticSettings(log2_hashmap_size=11), require_actor_grad=True,
                                        actor=ActorSettings(flip_prob=0.25))
        fcfg = NeuRADFieldConfig(grid=grid, **cfg)
        fld = NeuRADField(fcfg, actors=no_actors(), static_scale=100.0, implementation="torch").eval()
        fld.hashgrid.static_grid.hash_table.data = T(synth.hash_table(8 * 2**11, 4, seed=51, scale=0.5))
        for k, l in enumerate(fld.mlp_geo.layers):
            set_linear(l, 200 + 10 * k)
        for k, l in enumerate(fld.mlp_feature.layers):
            set_linear(l, 300 + 10 * k)
        R, S = 24, 12
        rb, (o, d, area, t) = make_bundle(R, seed=61)
        smp = PowerSampler(num_samples=S, lambda_=-1.0, scaling=0.1).eval()
        rs = smp(rb)
        out = fld(rs)
        starts, ends = rs.frustums.starts[..., 0], rs.frustums.ends[..., 0]
        g = rs.frustums.get_fast_isotropic_gaussian(1)
        kw = dict(o=o, d=d, area=area, starts=starts, ends=ends, feature=out[FieldHeadNames.FEATURE],
                  gmean=g.mean[:, :, 0], gstd=g.std[:, :, 0, 0])
        if cfg["use_sdf"]:
            kw.update(sdf=out[FieldHeadNames.SDF][..., 0], alpha=out[FieldHeadNames.ALPHA][..., 0])
        else:
            kw.update(density=out[FieldHeadNames.DENSITY][..., 0])
        # gradients through a random linear functional of the outputs (B1)
        gf = T(synth.normal(tuple(out[FieldHeadNames.FEATURE].shape), seed=71))
        key = FieldHeadNames.ALPHA if cfg["use_sdf"] else FieldHeadNames.DENSITY
        ga = T(synth.normal(tuple(out[key].shape), seed=72))
        ((out[FieldHeadNames.FEATURE] * gf).sum() + (out[key] * ga).sum()).backward()
        tg = fld.hashgrid.static_grid.hash_table.grad
        nz = tg.abs().sum(-1) > 0
        kw.update(g_feature=gf, g_head=ga[..., 0], tg_idx=nz.nonzero()[:, 0], tg_val=tg[nz],
                  **{f"geo_dw{k}": l.weight.grad for k, l in enumerate(fld.mlp_geo.layers)},
                  **{f"geo_db{k}": l.bias.grad for k, l in enumerate(fld.mlp_geo.layers)},
                  **{f"feat_dw{k}": l.weight.grad for k, l in enumerate(fld.mlp_feature.layers)},
                  **{f"feat_db{k}": l.bias.grad for k, l in enumerate(fld.mlp_feature.layers)})
        if cfg["use_sdf"]:
            kw.update(dbeta=fld.sdf_to_density.beta.grad)
        save(f"field_{tag}", **kw)


def make_prop(seed, lg=11):
    pcfg = NeuRADProposalFieldConfig()
    pcfg.grid.static.log2_hashmap_size = lg
    p = NeuRADProposalField(pcfg, actors=no_actors(), static_scale=100.0, implementation="torch").eval()
    p.hashgrid.static_grid.hash_table.data = T(synth.hash_table(6 * 2**lg, 1, seed=seed, scale=2.0))
    w, _ = synth.linear(1, 6, seed + 1, bias=False)
    p.density_decoder.weight.data = T(w + 0.3)
    return p


def golden_sampler():
    R = 20
    fars = synth.uniform((R,), 50.0, 30000.0, seed=83)
    rb, (o, d, area, t) = make_bundle(R, seed=81, fars=fars)
    props = [make_prop(91), make_prop(95)]
    # S1 power sampler
    ps = PowerSampler(num_samples=128, lambda_=-1.0, scaling=0.1).eval()
    rs0 = ps(rb)
    dens = props[1].get_density(rs0)[0]
    w0 = rs0.get_weights(dens)
    pdf = PDFSampler(include_original=False, single_jitter=True).eval()
    rs1 = pdf(rb, rs0, w0, num_samples=64)
    save("sampler_parts", o=o, d=d, area=area, fars=fars,
         sp0=torch.cat([rs0.spacing_starts[..., 0], rs0.spacing_ends[:, -1:, 0]], -1),
         eu0=torch.cat([rs0.frustums.starts[..., 0], rs0.frustums.ends[:, -1:, 0]], -1),
         dens0=dens[..., 0], w0=w0[..., 0],
         sp1=torch.cat([rs1.spacing_starts[..., 0], rs1.spacing_ends[:, -1:, 0]], -1),
         eu1=torch.cat([rs1.frustums.starts[..., 0], rs1.frustums.ends[:, -1:, 0]], -1))
    # S5+M1 the whole chain exactly as NeuRADModel drives it (incl. late-binding closure quirk, neurad.py:248)
    sampler = ProposalNetworkSampler(num_proposal_samples_per_ray=(128, 64), num_nerf_samples_per_ray=32,
                                     num_proposal_network_iterations=2, single_jitter=True,
                                     initial_sampler=PowerSampler(lambda_=-1.0, scaling=0.1),
                                     update_sched=lambda x: 0).eval()
    rb2, _ = make_bundle(R, seed=81, fars=fars)
    rb2.fars.clamp_max_(20000.0)
    density_fns = [lambda x: prop_field.get_density(x)[0] for prop_field in props]  # the reference's own idiom
    with torch.no_grad():
        rs, wl, rsl = sampler(rb2, density_fns, pass_ray_samples=True)
    save("sampler_chain", o=o, d=d, area=area, fars=fars, starts=rs.frustums.starts[..., 0],
         ends=rs.frustums.ends[..., 0], sps=rs.spacing_starts[..., 0], spe=rs.spacing_ends[..., 0],
         w0=wl[0][..., 0], w1=wl[1][..., 0], s0=rsl[0].frustums.starts[..., 0], e0=rsl[0].frustums.ends[..., 0],
         s1=rsl[1].frustums.starts[..., 0], e1=rsl[1].frustums.ends[..., 0])
    # training-mode jitter with injected randoms (t_rand / rand reproduced through torch.manual_seed)
    ps.train(), pdf.train()
    torch.manual_seed(7)
    t_rand = torch.rand((R, 129))
    torch.manual_seed(7)
    rs0t = ps(rb)
    torch.manual_seed(9)
    r1 = torch.rand((R, 1))
    torch.manual_seed(9)
    rs1t = pdf(rb, rs0t, w0, num_samples=64)
    save("sampler_train", fars=fars, t_rand=t_rand, rand1=r1, w0=w0[..., 0],
         sp0=torch.cat([rs0t.spacing_starts[..., 0], rs0t.spacing_ends[:, -1:, 0]], -1),
         eu0=torch.cat([rs0t.frustums.starts[..., 0], rs0t.frustums.ends[:, -1:, 0]], -1),
         sp1=torch.cat([rs1t.spacing_starts[..., 0], rs1t.spacing_ends[:, -1:, 0]], -1),
         eu1=torch.cat([rs1t.frustums.starts[..., 0], rs1t.frustums.ends[:, -1:, 0]], -1))
    # get_weights corner cases incl. the in-repo alpha-compositing sibling (cameras/rays.py:226-248)
    from nerfstudio.cameras.rays import RaySamples
    al = T(synth.uniform((16, 24, 1), 0.0, 1.0, seed=99))
    wa, tr = RaySamples.get_weights_and_transmittance_from_alphas(al)
    save("weights_alpha_eps", alphas=al[..., 0], w=wa[..., 0], trans=tr[..., 0])


if __name__ == "__main__":
    torch.manual_seed(0)
    torch.set_num_threads(8)
    golden_hashgrid()
    golden_sh_mlp_misc()
    golden_field()
    golden_sampler()
