"""Golden vectors for the model-level glue of the hot path, produced by the reference's own NeuRADModel (torch
implementation, CPU) -- run in the build container only:  python oracle/make_golden_model.py

  tests/golden/model_train_glue.npz   get_nff_outputs in TRAINING mode on a camera+lidar batch (models/neurad.py:368-421):
      appearance embedding (C3, :423-441), is_close_to_lidar / prop_weights_loss_i / non_nearby_* (C4, :399-419,677-700),
      depth / accumulation / weights_list; then the lidar terms of get_metrics_dict / get_loss_dict (:485-521,534-560)
      and decode_features' lidar head (:350-357) on those outputs.
      Deterministic: the samplers and fields are put in eval mode (no jitter, no actor flip) while the MODEL is in
      training mode; compositing uses the dense nerfacc 0.5.2 formulas (the reference substitutes a constant on CPU,
      models/neurad.py:713-715, and nerfacc is not installable here -- see oracle/neurad_oracle.py, C1).
  tests/golden/proposal_actors.npz    NeuRADProposalField.get_density with dynamic actors (fields/neurad_field.py:208-213):
      density and its autograd gradients w.r.t. the static table, the actor grids and the decoder.
The fixtures carry the model's state_dict, so the GPU test loads exactly these weights.
"""
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import ref_import

ref_import.install()
import synth  # noqa: E402
from make_golden import T, save  # noqa: E402
from make_golden_actors import trajectories  # noqa: E402
import nerfstudio.model_components.renderers as ref_renderers  # noqa: E402
import nerfstudio.models.neurad as ref_neurad  # noqa: E402
from nerfstudio.cameras.rays import RayBundle  # noqa: E402
from nerfstudio.data.scene_box import SceneBox  # noqa: E402
from nerfstudio.fields.neurad_field import NeuRADProposalField, NeuRADProposalFieldConfig  # noqa: E402
from nerfstudio.model_components.dynamic_actors import DynamicActors, DynamicActorsConfig  # noqa: E402
from nerfstudio.model_components.ray_samplers import PowerSampler  # noqa: E402


def dense_nerfacc():
    m = types.ModuleType("nerfacc")

    def render_weight_from_alpha(alphas, **kw):
        trans = torch.cumprod(torch.cat([torch.ones_like(alphas[..., :1]), 1 - alphas[..., :-1]], -1), -1)
        return trans * alphas, trans

    def accumulate_along_rays(weights, values=None, ray_indices=None, n_rays=None):
        return weights.sum(-1, keepdim=True) if values is None else (weights[..., None] * values).sum(-2)

    m.render_weight_from_alpha, m.accumulate_along_rays = render_weight_from_alpha, accumulate_along_rays
    return m


def fill(module, seed0):
    """deterministic, O(1)-feature parameters (tests/synth.py) for every tensor of the hot path"""
    for k, (name, p) in enumerate(module.named_parameters()):
        if name.endswith("hash_table"):
            p.data = T(synth.hash_table(p.shape[0], p.shape[1], seed=seed0 + k, scale=1.0 if p.shape[1] == 4 else 2.5))
        elif name.endswith("weight") and p.dim() == 2:
            w, _ = synth.linear(p.shape[0], p.shape[1], seed0 + k)
            p.data = T(w)
        elif name.endswith("bias"):
            p.data = T(synth.uniform(tuple(p.shape), -0.2, 0.2, seed0 + k))


def model_glue():
    ref_neurad.VGGPerceptualLossPix2Pix = torch.nn.Identity  # torchvision weights are not available
    na = dense_nerfacc()
    ref_neurad.nerfacc = na
    ref_renderers.nerfacc = na
    cfg = ref_neurad.NeuRADModelConfig(implementation="torch")
    cfg.field.grid.static.log2_hashmap_size = 10
    cfg.field.sdf_beta = 3.0
    for c in (cfg.field, cfg.sampling.proposal_field_1, cfg.sampling.proposal_field_2):
        c.grid.actor.use_4d_hashgrid = False
    for pf in (cfg.sampling.proposal_field_1, cfg.sampling.proposal_field_2):
        pf.grid.static.log2_hashmap_size = 9
    cfg.loss.vgg_mult = 0.0
    m = cfg.setup(scene_box=SceneBox(aabb=torch.tensor([[-100.0] * 3, [100.0] * 3])), num_train_data=2,
                  metadata={"duration": 8.0, "sensor_idx_to_name": {0: "cam0", 1: "cam1", 2: "lidar"},
                            "trajectories": []})
    type(m)._render_weights = lambda self, outputs, rs: na.render_weight_from_alpha(
        outputs[ref_neurad.FieldHeadNames.ALPHA].squeeze(-1))[0]
    fill(m.field, 100), fill(m.proposal_fields[0], 200), fill(m.proposal_fields[1], 300), fill(m.lidar_decoder, 400)
    m.appearance_embedding.weight.data = T(synth.normal(tuple(m.appearance_embedding.weight.shape), seed=77))
    m.train()
    m.sampler.eval(), m.field.eval()
    for p in m.proposal_fields:
        p.eval()
    # batch: 32 camera rays (2 sensors) then 48 lidar rays (with and without a return)
    Rc, Rl = 32, 48
    R = Rc + Rl
    o = synth.normal((R, 3), 5) * np.array([4.0, 4.0, 0.5], np.float32)
    d = synth.normal((R, 3), 6)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    is_lidar = np.arange(R) >= Rc
    did_return = np.where(is_lidar, synth.uniform((R,), 0, 1, 7) < 0.75, True)
    dist = synth.uniform((R,), 2.0, 80.0, 8)
    area = np.where(is_lidar, 4.5e-6, 2.7e-7).astype(np.float32)
    times = synth.uniform((R,), 0.0, 8.0, 9)
    sensor = np.where(is_lidar, 2, np.arange(R) % 2).astype(np.int64)

    def bundle():
        return RayBundle(origins=T(o), directions=T(d.astype(np.float32)), pixel_area=T(area)[:, None],
                         times=T(times)[:, None],
                         metadata={"is_lidar": torch.from_numpy(is_lidar)[:, None],
                                   "did_return": torch.from_numpy(did_return)[:, None],
                                   "directions_norm": T(dist)[:, None], "sensor_idxs": torch.from_numpy(sensor)[:, None]})

    out = m.get_nff_outputs(bundle(), calc_lidar_losses=True)
    gold = {"o": o, "d": d.astype(np.float32), "area": area, "times": times, "is_lidar": is_lidar,
            "did_return": did_return, "directions_norm": dist, "sensor_idxs": sensor, "duration": 8.0,
            "features": out["features"], "depth": out["depth"], "accumulation": out["accumulation"]}
    for i in range(2):
        gold[f"prop_depth_{i}"] = out[f"prop_depth_{i}"]
        gold[f"prop_weights_loss_{i}"] = out[f"prop_weights_loss_{i}"]
        gold[f"weights_{i}"] = out["weights_list"][i][..., 0]
        gold[f"close_{i}"] = out["ray_samples_list"][i].metadata["is_close_to_lidar"][..., 0]
    gold["weights_2"] = out["weights_list"][2][..., 0]
    gold["close_2"] = out["ray_samples_list"][2].metadata["is_close_to_lidar"][..., 0]
    gold["non_nearby_weights"] = out["non_nearby_weights"]
    gold["non_nearby_lidar_ray_indices"] = out["non_nearby_lidar_ray_indices"]
    # ---- lidar head + lidar losses on these outputs (get_outputs :334-348, get_metrics_dict :485-521) ----------
    rgb, intensity, ray_drop_logits = None, None, None
    lidar_feat = out["features"][torch.from_numpy(is_lidar)]
    intensity, ray_drop_logits = m.lidar_decoder(lidar_feat).split(1, dim=-1)
    intensity = intensity.sigmoid()
    outputs = dict(out)
    outputs["intensity"], outputs["ray_drop_logits"] = intensity, ray_drop_logits
    batch = {"lidar": T(np.concatenate([synth.normal((Rl, 3), 21), synth.uniform((Rl, 1), 0, 1, 22)], -1)),
             "is_lidar": torch.from_numpy(is_lidar)[:, None], "did_return": torch.from_numpy(did_return)[:, None],
             "distance": T(dist[is_lidar])[:, None]}
    metrics = m.get_metrics_dict(outputs, batch)
    losses = m.get_loss_dict(outputs, batch, metrics)
    gold["intensity"], gold["ray_drop_logits"] = intensity, ray_drop_logits
    gold["lidar_points"] = batch["lidar"]
    for k in ("depth_loss", "intensity_loss", "ray_drop_loss", "carving_loss", "depth_loss_0", "depth_loss_1",
              "carving_loss_0", "carving_loss_1", "distortion"):
        gold["metric_" + k] = metrics[k]
    for k in ("interlevel_loss", "distortion_loss", "depth_loss", "intensity_loss", "carving_loss", "ray_drop_loss",
              "depth_loss_0", "carving_loss_0", "depth_loss_1", "carving_loss_1"):
        gold["loss_" + k] = losses[k]
    # gradient of the summed lidar losses w.r.t. the rendered lidar depth and the final weights' table (end to end)
    total = sum(v for k, v in losses.items())
    m.zero_grad()
    total.backward()
    gold["g_field_table_abs_sum"] = m.field.hashgrid.static_grid.hash_table.grad.abs().sum()
    gold["g_prop1_table_abs_sum"] = m.proposal_fields[1].hashgrid.static_grid.hash_table.grad.abs().sum()
    # the two table gradients themselves, sparse (flat element index, value): a sign error in half the rows passes an
    # abs-sum comparison, not an element-wise one
    for key, t in (("g_field_table", m.field.hashgrid.static_grid.hash_table.grad),
                   ("g_prop1_table", m.proposal_fields[1].hashgrid.static_grid.hash_table.grad)):
        flat = t.reshape(-1)
        nz = flat.nonzero()[:, 0]
        gold[key + "_idx"], gold[key + "_val"] = nz, flat[nz]
    gold["g_lidar_decoder_w0"] = m.lidar_decoder.layers[0].weight.grad
    gold["g_embedding"] = m.appearance_embedding.weight.grad
    gold["g_beta"] = m.field.sdf_to_density.beta.grad
    lc = cfg.loss
    gold["loss_cfg"] = np.array([lc.depth_mult, lc.intensity_mult, lc.carving_mult, lc.ray_drop_loss_mult,
                                 lc.prop_lidar_loss_mult, lc.non_return_loss_mult, lc.non_return_lidar_distance,
                                 lc.quantile_threshold, lc.carving_epsilon, lc.interlevel_loss_mult,
                                 lc.distortion_loss_mult], np.float64)
    for name, t in m.state_dict().items():
        if name.split(".")[0] in ("field", "proposal_fields", "appearance_embedding", "lidar_decoder"):
            gold["sd/" + name] = t
    save("model_train_glue", **gold)


def proposal_with_actors():
    actors = DynamicActors(DynamicActorsConfig(), trajectories=trajectories())
    cfg = NeuRADProposalFieldConfig()
    cfg.grid.static.log2_hashmap_size = 10
    cfg.grid.actor.log2_hashmap_size = 8
    cfg.grid.actor.use_4d_hashgrid = False
    fld = NeuRADProposalField(cfg, actors=actors, static_scale=100.0, implementation="torch").eval()
    actors.eval()
    fld.hashgrid.static_grid.hash_table.data = T(synth.hash_table(6 * 2**10, 1, seed=61, scale=2.0))
    for i, g in enumerate(fld.hashgrid.actor_grids):
        g.hash_table.data = T(synth.hash_table(4 * 2**8, 1, seed=500 + i, scale=2.5))
    fld.density_decoder.weight.data = T(synth.uniform((1, 6), -0.6, 0.6, seed=62))
    R, S = 48, 40  # the rays of make_golden_actors.py: aimed at the actors' corridor
    o = synth.normal((R, 3), 7) * np.array([1.0, 1.0, 0.2], np.float32)
    tgt = np.stack([synth.uniform((R,), 10, 24, 8), np.where(np.arange(R) % 2 == 0, 8.0, -5.5)
                    + synth.uniform((R,), -1.5, 1.5, 9), synth.uniform((R,), 0.0, 1.0, 10)], -1).astype(np.float32)
    d = tgt - o
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    times = synth.uniform((R,), -0.5, 4.5, 11)
    rb = RayBundle(origins=T(o), directions=T(d.astype(np.float32)), pixel_area=torch.full((R, 1), 2.43e-6),
                   times=T(times)[:, None], nears=torch.zeros(R, 1), fars=torch.full((R, 1), 60.0))
    rs = PowerSampler(num_samples=S, lambda_=-1.0, scaling=0.1).eval()(rb)
    dens, _ = fld.get_density(rs)
    g = T(synth.normal((R, S, 1), seed=73))
    (dens * g).sum().backward()
    tg = fld.hashgrid.static_grid.hash_table.grad
    nz = tg.abs().sum(-1) > 0
    gold = dict(o=o, d=d.astype(np.float32), times=times, starts=rs.frustums.starts[..., 0], ends=rs.frustums.ends[..., 0],
                density=dens[..., 0], g_density=g[..., 0], tg_idx=nz.nonzero()[:, 0], tg_val=tg[nz],
                g_decoder=fld.density_decoder.weight.grad,
                dpos_is_none=np.array(actors.actor_positions.grad is None))
    for i, gr in enumerate(fld.hashgrid.actor_grids):
        gold[f"ag{i}"] = gr.hash_table.grad if gr.hash_table.grad is not None else torch.zeros_like(gr.hash_table)
    save("proposal_actors", **gold)


def cnn_decoder():
    """the reference's rgb_decoder (models/neurad.py:198-216) on two 8x8 feature patches, training (batch-statistics BN)
    and eval mode"""
    ref_neurad.VGGPerceptualLossPix2Pix = torch.nn.Identity
    cfg = ref_neurad.NeuRADModelConfig(implementation="torch")
    cfg.field.grid.static.log2_hashmap_size = 8
    for c in (cfg.field, cfg.sampling.proposal_field_1, cfg.sampling.proposal_field_2):
        c.grid.actor.use_4d_hashgrid = False
        c.grid.static.log2_hashmap_size = 8
    m = cfg.setup(scene_box=SceneBox(aabb=torch.tensor([[-100.0] * 3, [100.0] * 3])), num_train_data=2,
                  metadata={"duration": 8.0, "sensor_idx_to_name": {0: "cam0"}, "trajectories": []})
    dec = m.rgb_decoder
    for k, (name, p) in enumerate(dec.named_parameters()):
        fan_in = p[0].numel() if p.dim() > 1 else 1
        p.data = T((synth.normal(tuple(p.shape), 600 + k) * (1.0 / np.sqrt(fan_in) if p.dim() > 1 else 0.1)).astype(np.float32))
        if name.endswith(("1.weight", "4.weight")) and p.dim() == 1:  # BatchNorm scale around 1
            p.data = p.data + 1.0
    feats = T(synth.normal((2 * 8 * 8, 48), 650))
    gold = {"features": feats}
    dec.train()
    rgb, _, _ = m.decode_features(feats, patch_size=(8, 8))
    gold["rgb_train"] = rgb
    gold["bn_running_mean"] = dec[2].main_branch[1].running_mean
    dec.eval()
    rgb, _, _ = m.decode_features(feats, patch_size=(8, 8))
    gold["rgb_eval"] = rgb
    gold["param_names"] = np.array([n for n, _ in dec.named_parameters()])  # weights are re-derived from tests/synth.py
    save("cnn_decoder", **gold)


if __name__ == "__main__":
    torch.manual_seed(0)
    cnn_decoder()
    proposal_with_actors()
    model_glue()
