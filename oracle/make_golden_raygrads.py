"""tests/golden/ray_grads.npz: dL/d(origins, directions) of the hash-grid path, by RUNNING THE REFERENCE's autograd.

A camera optimizer that moves the rays (cameras/camera_optimizers.py:173-182, mode SO3xR3: the ``*-scaleopt`` methods,
configs/method_configs.py:438-447) gets its rendering-loss gradient through exactly one chain of the hot path:
mean = origins + directions * t (cameras/rays.py:109-124; t comes from detached bins, ray_samplers.py:363-364) ->
ScaledSceneContraction of the GaussiansStd (spatial_distortions.py:126-141: the contracted std depends on |mean|_inf too) ->
HashEncoding.pytorch_fwd's offsets (encodings.py:425-464) and the std-dependent feature rescale (neurad_encoding.py:297-304).
View directions carry none: SHEncoding.pytorch_fwd is @torch.no_grad (encodings.py:797).

Three cases on the same rays: the main field's static encoding alone (a random linear functional of the rescaled
features), a proposal field's density, and the whole NeuRADField forward (feature + alpha functional of golden field_sdf).
Build container only:  python oracle/make_golden_raygrads.py
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
from make_golden import T, make_prop, no_actors, save, set_linear  # noqa: E402
from nerfstudio.cameras.rays import RayBundle  # noqa: E402
from nerfstudio.field_components.field_heads import FieldHeadNames  # noqa: E402
from nerfstudio.field_components.neurad_encoding import (  # noqa: E402
    ActorSettings, NeuRADHashEncodingConfig, StaticSettings)
from nerfstudio.fields.neurad_field import NeuRADField, NeuRADFieldConfig  # noqa: E402
from nerfstudio.model_components.ray_samplers import PowerSampler  # noqa: E402


def bundle(R, seed):
    """rays whose samples lie on both sides of the contraction's unit cube (static_scale = 100 m), a few of them
    axis-aligned so that the inf-norm's arg-max coordinate changes along the ray"""
    o, d, area, t = synth.rays(R, seed)
    d[:3] = np.array([[1, 0.02, 0.01], [0.03, -1, 0.2], [0.5, 0.5, 0.7071]], np.float32)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    ot = T(o).requires_grad_(True)
    dt = T(d).requires_grad_(True)
    rb = RayBundle(origins=ot, directions=dt, pixel_area=T(area)[:, None], times=T(t)[:, None],
                   nears=torch.zeros(R, 1), fars=torch.full((R, 1), 20000.0))
    return rb, ot, dt, (o, d, area)


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    R, S = 40, 24
    grid = NeuRADHashEncodingConfig(static=StaticSettings(log2_hashmap_size=11), require_actor_grad=True,
                                    actor=ActorSettings(flip_prob=0.25))
    fld = NeuRADField(NeuRADFieldConfig(grid=grid, use_sdf=True), actors=no_actors(), static_scale=100.0,
                      implementation="torch").eval()
    fld.hashgrid.static_grid.hash_table.data = T(synth.hash_table(8 * 2**11, 4, seed=51, scale=0.5))
    for k, l in enumerate(fld.mlp_geo.layers):
        set_linear(l, 200 + 10 * k)
    for k, l in enumerate(fld.mlp_feature.layers):
        set_linear(l, 300 + 10 * k)
    smp = PowerSampler(num_samples=S, lambda_=-1.0, scaling=0.1).eval()
    kw = {}

    # (1) the static encoding alone
    rb, ot, dt, (o, d, area) = bundle(R, 161)
    rs = smp(rb)
    starts, ends = rs.frustums.starts[..., 0].detach(), rs.frustums.ends[..., 0].detach()
    feats, _ = fld.hashgrid(rs.frustums.get_fast_isotropic_gaussian(1), rs.times, rs.frustums.directions)
    g_enc = T(synth.normal(tuple(feats.shape), seed=171))
    (feats * g_enc).sum().backward()
    kw.update(o=o, d=d, area=area, starts=starts, ends=ends, g_enc=g_enc, enc_go=ot.grad.clone(), enc_gd=dt.grad.clone())

    # (2) a proposal field's density (fields/neurad_field.py:208-213)
    prop = make_prop(91)
    rb, ot, dt, _ = bundle(R, 161)
    rs = smp(rb)
    dens = prop.get_density(rs)[0]
    g_dens = T(synth.normal(tuple(dens.shape), seed=173))
    (dens * g_dens).sum().backward()
    kw.update(prop_g_dens=g_dens[..., 0], prop_dens=dens[..., 0].detach(), prop_go=ot.grad.clone(), prop_gd=dt.grad.clone())

    # (3) the whole field forward (fields/neurad_field.py:128-152), the functional of golden field_sdf
    rb, ot, dt, _ = bundle(R, 161)
    rs = smp(rb)
    out = fld(rs)
    gf = T(synth.normal(tuple(out[FieldHeadNames.FEATURE].shape), seed=71))
    ga = T(synth.normal(tuple(out[FieldHeadNames.ALPHA].shape), seed=72))
    ((out[FieldHeadNames.FEATURE] * gf).sum() + (out[FieldHeadNames.ALPHA] * ga).sum()).backward()
    kw.update(field_g_feature=gf, field_g_alpha=ga[..., 0], field_go=ot.grad.clone(), field_gd=dt.grad.clone(),
              field_feature=out[FieldHeadNames.FEATURE].detach())
    # the same functional with the reference in fp64: its own fp32 noise floor on these gradients (a hidden unit of the
    # MLPs within rounding of the ReLU kink flips between the precisions and switches one sample's contribution; the
    # direction gradient weighs every sample with its distance t, up to 2e4 m)
    fld64 = fld.double()
    rb, ot64, dt64, _ = bundle(R, 161)
    rb.origins, rb.directions = ot64.double(), dt64.double()
    rb.pixel_area, rb.times, rb.nears, rb.fars = rb.pixel_area.double(), rb.times.double(), rb.nears.double(), rb.fars.double()
    out = fld64(smp(rb))
    ((out[FieldHeadNames.FEATURE] * gf.double()).sum() + (out[FieldHeadNames.ALPHA] * ga.double()).sum()).backward()
    rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())  # noqa: E731
    per_ray = ((kw["field_gd"].double() - dt64.grad.double()).norm(dim=-1) / dt64.grad.double().norm(dim=-1))
    kw.update(field_floor=np.array([rel(kw["field_go"], ot64.grad), rel(kw["field_gd"], dt64.grad)]),
              field_gd_rays_off_fp64=np.array(int((per_ray > 1e-4).sum())))
    print("fp32 vs fp64 floor of the field's ray gradients (origins, directions):", kw["field_floor"],
          "rays further than 1e-4:", int((per_ray > 1e-4).sum()))
    save("ray_grads", **kw)


if __name__ == "__main__":
    main()
