"""Golden vectors for the dynamic-actor path, produced by the reference itself
(NeuRADField(implementation="torch") with a DynamicActors holding 3 synthetic trajectories).  Run in the build
container only:  python oracle/make_golden_actors.py"""
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
from make_golden import T, make_bundle, save, set_linear  # noqa: E402
from nerfstudio.field_components.field_heads import FieldHeadNames  # noqa: E402
from nerfstudio.field_components.neurad_encoding import ActorSettings, NeuRADHashEncodingConfig, StaticSettings  # noqa: E402
from nerfstudio.fields.neurad_field import NeuRADField, NeuRADFieldConfig  # noqa: E402
from nerfstudio.model_components.dynamic_actors import DynamicActors, DynamicActorsConfig  # noqa: E402
from nerfstudio.model_components.ray_samplers import PowerSampler  # noqa: E402


def trajectories():
    """3 actors moving along +x at different offsets; actor 2 overlaps actor 1's box; actor 0 only present early."""
    ts_all = torch.tensor([0.0, 1.0, 2.0, 3.0, 4.0])
    out = []
    for a, (y0, yaw, dims, ts) in enumerate([(8.0, 0.3, (2.0, 4.5, 1.6), ts_all[:3]), (-6.0, -0.2, (2.1, 4.8, 1.7), ts_all),
                                             (-5.0, 0.1, (1.9, 4.2, 1.5), ts_all[1:])]):
        poses = []
        for t in ts:
            c, s = np.cos(yaw + 0.05 * float(t)), np.sin(yaw + 0.05 * float(t))
            p = torch.eye(4)
            p[:3, :3] = torch.tensor([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])
            p[:3, 3] = torch.tensor([12.0 + 2.0 * float(t) + a, y0, 0.5])
            poses.append(p)
        out.append({"timestamps": ts.clone(), "poses": torch.stack(poses), "dims": torch.tensor(dims),
                    "symmetric": torch.tensor(True), "deformable": torch.tensor(False)})
    return out


def main():
    actors = DynamicActors(DynamicActorsConfig(), trajectories=trajectories())
    grid = NeuRADHashEncodingConfig(static=StaticSettings(log2_hashmap_size=11),
                                    actor=ActorSettings(flip_prob=0.25, log2_hashmap_size=9, use_4d_hashgrid=False))
    fld = NeuRADField(NeuRADFieldConfig(grid=grid), actors=actors, static_scale=100.0, implementation="torch").eval()
    actors.eval()
    fld.hashgrid.static_grid.hash_table.data = T(synth.hash_table(8 * 2**11, 4, seed=51, scale=0.5))
    for i, g in enumerate(fld.hashgrid.actor_grids):
        g.hash_table.data = T(synth.hash_table(4 * 2**9, 4, seed=400 + i, scale=0.7))
    for k, l in enumerate(fld.mlp_geo.layers):
        set_linear(l, 200 + 10 * k)
    for k, l in enumerate(fld.mlp_feature.layers):
        set_linear(l, 300 + 10 * k)
    # rays from the origin region aimed at the actors' corridor so that many samples fall inside boxes
    R, S = 48, 40
    o = synth.normal((R, 3), 7) * np.array([1.0, 1.0, 0.2], np.float32)
    tgt = np.stack([synth.uniform((R,), 10, 24, 8), np.where(np.arange(R) % 2 == 0, 8.0, -5.5)
                    + synth.uniform((R,), -1.5, 1.5, 9), synth.uniform((R,), 0.0, 1.0, 10)], -1).astype(np.float32)
    d = tgt - o
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    times = synth.uniform((R,), -0.5, 4.5, 11)
    from nerfstudio.cameras.rays import RayBundle
    rb = RayBundle(origins=T(o), directions=T(d.astype(np.float32)), pixel_area=torch.full((R, 1), 2.43e-6),
                   times=T(times)[:, None], nears=torch.zeros(R, 1), fars=torch.full((R, 1), 60.0))
    rs = PowerSampler(num_samples=S, lambda_=-1.0, scaling=0.1).eval()(rb)
    with torch.no_grad():
        out = fld(rs)
        gauss = rs.frustums.get_fast_isotropic_gaussian(1)
        feats, dirs = fld.hashgrid(gauss, rs.times, rs.frustums.directions)
        b2w, valid = actors.get_boxes2world(rs.times[:, 0].squeeze(-1), flatten=False)
        idx, _, _ = fld.hashgrid._split_static_vs_actors(gauss, rs.times, rs.frustums.directions)
    print("actor-hit samples:", idx[0].shape[0], "of", R * S)
    # B1 with actors: gradients w.r.t. static table, actor grids and actor trajectories (require_actor_grad=True)
    out_g = fld(rs)
    gf = T(synth.normal(tuple(out_g[FieldHeadNames.FEATURE].shape), seed=71))
    ga = T(synth.normal(tuple(out_g[FieldHeadNames.ALPHA].shape), seed=72))
    ((out_g[FieldHeadNames.FEATURE] * gf).sum() + (out_g[FieldHeadNames.ALPHA] * ga).sum()).backward()
    tg = fld.hashgrid.static_grid.hash_table.grad
    nz = tg.abs().sum(-1) > 0
    grads = dict(g_feature=gf, g_alpha=ga[..., 0], tg_idx=nz.nonzero()[:, 0], tg_val=tg[nz],
                 dpos=actors.actor_positions.grad, drot=actors.actor_rotations_6d.grad)
    for i, g in enumerate(fld.hashgrid.actor_grids):
        grads[f"ag{i}"] = g.hash_table.grad if g.hash_table.grad is not None else torch.zeros_like(g.hash_table)
    # dL/dx of a plain HashEncoding (autograd of encodings.py:425-464 w.r.t. in_tensor)
    from nerfstudio.field_components.encodings import HashEncoding
    enc = HashEncoding(num_levels=4, min_res=64, max_res=1024, log2_hashmap_size=9, features_per_level=4,
                       implementation="torch")
    enc.hash_table.data = T(synth.hash_table(4 * 2**9, 4, seed=400, scale=0.7))
    x = T(synth.uniform((200, 3), 0.0, 1.0, seed=3)).requires_grad_(True)
    y = enc(x)
    gy = T(synth.normal(tuple(y.shape), seed=5))
    (y * gy).sum().backward()
    grads.update(hx=x.detach(), hgy=gy, hdx=x.grad)
    save("field_actors_grads", **grads)
    save("field_actors", o=o, d=d.astype(np.float32), area=np.full((R,), 2.43e-6, np.float32), times=times,
         starts=rs.frustums.starts[..., 0], ends=rs.frustums.ends[..., 0], feature=out[FieldHeadNames.FEATURE],
         sdf=out[FieldHeadNames.SDF][..., 0], alpha=out[FieldHeadNames.ALPHA][..., 0], enc=feats, directions=dirs,
         b2w=b2w, valid=valid, hit_ray=idx[0], hit_sample=idx[1], hit_actor=idx[2],
         timestamps=actors.unique_timestamps, positions=actors.actor_positions, rotations_6d=actors.actor_rotations_6d,
         present=actors.actor_present_at_time, sizes=actors.actor_sizes, padding=actors.actor_padding)


if __name__ == "__main__":
    torch.manual_seed(0)
    main()
