"""GPU tests of the training step's glue kernels (csrc/train_fused.hip) and of the fused training path built on them
(models/neurad.py:_fused_train_nff_outputs): every kernel against plain torch fp32/fp64 autograd of the reference's formula,
the composed step against the operator-level path of the same model and against the reference's own outputs / gradients
(tests/golden/model_train_glue.npz)."""
import numpy as np
import pytest
import torch

import synth
from conftest import load_golden, rel_l2

pytestmark = pytest.mark.gpu
TOL = 1e-4
TIGHT = 2e-5


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from neurad_studio_amd import ops as _ops

    return _ops


def dev(a, dtype=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda", dtype)


def host(t):
    return t.detach().cpu().numpy()


def _edges(R, S, seed, extra=0):
    e = np.sort(synth.uniform((R, S + 1 + extra), 0.0, 60.0, seed=seed), -1)
    return dev(e)


@pytest.mark.parametrize("shape", [(19, 70), (5, 128), (33, 64), (1, 3)])
def test_prop_weights_from_edges_fwd_bwd_vs_torch(ops, shape):
    """S3 (cameras/rays.py:188-210) + render_depth_simple (models/neurad.py:727-734) from bin edges, any row stride"""
    R, S = shape
    big = _edges(R, S, 3, extra=5)  # a wider tensor: the kernel sees a row stride of S + 6
    edges = big[:, :S + 1]
    dens = dev(synth.uniform((R, S), 0.0, 0.4, seed=4))
    w, depth = ops.prop_weights_fwd(edges, dens)
    e64 = edges.double()
    td = dens.double().clone().requires_grad_(True)
    sd = (e64[:, 1:] - e64[:, :-1]) * td
    tr = torch.exp(-torch.cat([torch.zeros(R, 1, dtype=torch.float64, device="cuda"), torch.cumsum(sd[:, :-1], -1)], -1))
    tw = (1 - torch.exp(-sd)) * tr
    tdepth = (tw * (e64[:, 1:] + e64[:, :-1]) / 2).sum(-1, keepdim=True)
    assert rel_l2(host(w), host(tw)) < TIGHT and rel_l2(host(depth), host(tdepth)) < TIGHT
    # the kernel the operator-level path uses, on materialised deltas: same numbers
    w_old = ops.weights_from_density((edges[:, 1:] - edges[:, :-1]).contiguous(), dens)
    assert torch.equal(w, w_old)
    gw, gd = dev(synth.normal((R, S), 5)), dev(synth.normal((R, 1), 6))
    ((tw * gw.double()).sum() + (tdepth * gd.double()).sum()).backward()
    assert rel_l2(host(ops.prop_weights_bwd(edges, dens, gw, gd)), host(td.grad)) < TOL
    td.grad = None
    sd = (e64[:, 1:] - e64[:, :-1]) * td
    tr = torch.exp(-torch.cat([torch.zeros(R, 1, dtype=torch.float64, device="cuda"), torch.cumsum(sd[:, :-1], -1)], -1))
    ((1 - torch.exp(-sd)) * tr * gw.double()).sum().backward()
    assert rel_l2(host(ops.prop_weights_bwd(edges, dens, gw, None)), host(td.grad)) < TOL


def _torch_sdf_render(sdf, beta, beta_min, feat, edges):
    """models/neurad.py:373-395 + model_components/utils.py:21-41 as torch ops (fp64)"""
    R, S = sdf.shape
    alpha = torch.sigmoid(-sdf * (beta.abs() + beta_min))
    trans = torch.cumprod(torch.cat([torch.ones(R, 1, dtype=sdf.dtype, device=sdf.device), 1 - alpha[:, :-1]], -1), -1)
    w = alpha * trans
    acc = w.sum(-1, keepdim=True)
    w2 = torch.cat([w[:, :-1], w[:, -1:] + 1 - acc], -1)
    out = (w2[..., None] * feat).sum(1)
    mid = (edges[:, :-1] + edges[:, 1:]) / 2
    depth = (w2[:, :-1] * mid[:, :-1]).sum(-1, keepdim=True)
    return alpha, w2[:, :-1], out, depth, acc


@pytest.mark.parametrize("pair", ["0", "1"], ids=["ray-per-wave", "two-rays-per-wave"])
@pytest.mark.parametrize("cfg", [(23, 32, 32, 16, 2.5), (7, 70, 32, 0, -3.0), (9, 5, 12, 4, 0.7), (4, 2, 3, 0, 20.0),
                                 (10, 17, 32, 0, 1.5), (1, 32, 32, 0, 4.0)])
def test_sdf_render_fwd_bwd_vs_torch(ops, cfg, pair, switches):
    """pair = "1": NRHIP_SDF_RENDER_PAIR, the kernels with one ray per 32-lane half (taken for S <= 32 and 32 channels; the
    odd ray counts leave a half without a ray)"""
    switches.set("NRHIP_SDF_RENDER_PAIR", pair)
    R, S, Cc, A, b = cfg
    sdf = dev(synth.normal((R, S), 1) * 0.5)
    feat = dev(synth.normal((R, S, Cc), 2))
    edges = _edges(R, S, 3)
    beta = torch.tensor([b], device="cuda")
    alpha, w_ns, out, depth, acc = ops.sdf_render_fwd(sdf, beta, 1e-4, feat, edges, extra_cols=A)
    assert out.shape == (R, Cc + A)
    ts, tf = sdf.double().requires_grad_(True), feat.double().requires_grad_(True)
    tb = beta.double().requires_grad_(True)
    ra, rw, ro, rd, rc = _torch_sdf_render(ts, tb, 1e-4, tf, edges.double())
    assert rel_l2(host(alpha), host(ra)) < TIGHT and rel_l2(host(w_ns), host(rw)) < TIGHT
    assert rel_l2(host(out[:, :Cc]), host(ro)) < TIGHT and rel_l2(host(depth), host(rd)) < TIGHT
    assert rel_l2(host(acc), host(rc)) < TIGHT
    gF = dev(synth.normal((R, Cc + A), 7))
    gD, gA, gW = dev(synth.normal((R, 1), 8)), dev(synth.normal((R, 1), 9)), dev(synth.normal((R, S - 1), 10))
    ((ro * gF[:, :Cc].double()).sum() + (rd * gD.double()).sum() + (rc * gA.double()).sum()
     + (rw * gW.double()).sum()).backward()
    gfeat, gsdf, gbeta = ops.sdf_render_bwd(sdf, beta, 1e-4, alpha, feat, edges, gF[:, :Cc], gD, gA, gW)
    assert rel_l2(host(gfeat), host(tf.grad)) < TIGHT
    assert rel_l2(host(gsdf), host(ts.grad)) < TOL
    assert abs(float(gbeta) / float(tb.grad) - 1) < 1e-3
    # optional upstream gradients absent
    ts.grad = tf.grad = tb.grad = None
    ra, rw, ro, rd, rc = _torch_sdf_render(ts, tb, 1e-4, tf, edges.double())
    (ro * gF[:, :Cc].double()).sum().backward()
    gfeat, gsdf, gbeta = ops.sdf_render_bwd(sdf, beta, 1e-4, alpha, feat, edges, gF[:, :Cc], None, None, None)
    assert rel_l2(host(gsdf), host(ts.grad)) < TOL and abs(float(gbeta) / float(tb.grad) - 1) < 1e-3
    # bit-reproducible (fixed summation order of d beta)
    again = ops.sdf_render_bwd(sdf, beta, 1e-4, alpha, feat, edges, gF[:, :Cc], None, None, None)
    assert torch.equal(again[2], gbeta) and torch.equal(again[1], gsdf)


@pytest.mark.parametrize("temporal", [True, False])
def test_appearance_kernels_vs_reference_formula(ops, temporal):
    """models/neurad.py:423-441 with the slot arithmetic in the kernel; writes into a column block of a wider row"""
    R, D, n_sensors, n_per, duration = 1000, 16, 7, 8, 8.0
    torch.manual_seed(3)
    table = torch.randn(n_sensors * (n_per if temporal else 1), D, device="cuda")
    sensor = torch.randint(0, n_sensors, (R, 1), device="cuda")
    times = torch.rand(R, 1, device="cuda") * duration
    times[:5] = torch.tensor([0.0, duration, duration / 2, 7.999, 1.0], device="cuda")[:, None]
    out = torch.zeros(R, 32 + D, device="cuda")
    ops.appearance_fwd(table, sensor, times if temporal else None, duration, n_per, temporal, R, out=out[:, 32:])
    tt = table.clone().requires_grad_(True)
    if temporal:
        ti = times / duration * n_per
        lo = ti.floor().clamp(0, n_per - 1)
        hi = (lo + 1).clamp(0, n_per - 1)
        fr = ti - lo
        ref = tt[(lo + sensor * n_per).squeeze(-1).long()] * (1 - fr) + tt[(hi + sensor * n_per).squeeze(-1).long()] * fr
    else:
        ref = tt[sensor.squeeze(-1)]
    assert float(out[:, :32].abs().max()) == 0.0
    assert rel_l2(host(out[:, 32:]), host(ref)) < 1e-6
    g = torch.randn(R, 32 + D, device="cuda")
    (ref * g[:, 32:]).sum().backward()
    gw = ops.appearance_bwd(g[:, 32:], sensor, times if temporal else None, duration, n_per, temporal, table.shape[0])
    assert rel_l2(host(gw), host(tt.grad)) < 1e-5


def test_mask_compact_matches_nonzero(ops):
    for R, p in ((57344, 0.3), (1000, 0.0), (1025, 1.0), (3, 0.5)):
        m = torch.rand(R, device="cuda") < p
        n = int(m.sum())
        rows, inv = ops.mask_compact(m, n)
        assert torch.equal(rows, m.nonzero().squeeze(-1))
        ref_inv = torch.full((R,), -1, dtype=torch.int32, device="cuda")
        ref_inv[m] = torch.arange(n, dtype=torch.int32, device="cuda")
        assert torch.equal(inv, ref_inv)


@pytest.mark.parametrize("n_lidar", [16384, 777, 2])
def test_lidar_losses_kernel_vs_torch_formulation(ops, n_lidar):
    """nrhip_lidar_losses (one launch, radix-select quantile) against the torch formulation of models/neurad.py:485-521
    (lidar_metrics(fused=False): torch.quantile, masked means, BCE): values and gradients"""
    from neurad_studio_amd.model_components.lidar_losses import LidarLossSettings, lidar_metrics

    torch.manual_seed(5)
    R = n_lidar * 3 + 11
    is_lidar = torch.zeros(R, dtype=torch.bool, device="cuda")
    is_lidar[torch.randperm(R, device="cuda")[:n_lidar]] = True
    did_return = torch.rand(n_lidar, device="cuda") < 0.8
    if n_lidar == 2:
        did_return[:] = True
    distance = torch.rand(n_lidar, 1, device="cuda") * 78 + 2
    target = torch.rand(n_lidar, 1, device="cuda")
    cfg = LidarLossSettings()

    def leaves():
        torch.manual_seed(6)
        d = [(torch.rand(R, 1, device="cuda") * 120 + 1).requires_grad_(True) for _ in range(3)]
        i = torch.rand(n_lidar, 1, device="cuda").requires_grad_(True)
        lg = torch.randn(n_lidar, 1, device="cuda").requires_grad_(True)
        return d, i, lg

    res = []
    for fused in (True, False):
        d, i, lg = leaves()
        outputs = {"depth": d[0], "prop_depth_0": d[1], "prop_depth_1": d[2], "intensity": i, "ray_drop_logits": lg,
                   "non_nearby_weights_loss": torch.tensor(3.0, device="cuda"),
                   "prop_weights_loss_0": torch.tensor(1.0, device="cuda"),
                   "prop_weights_loss_1": torch.tensor(2.0, device="cuda")}
        m = lidar_metrics(outputs, is_lidar, did_return, distance, target, cfg, fused=fused)
        coef = {"depth_loss": 1.0, "intensity_loss": 0.7, "ray_drop_loss": 1.3, "depth_loss_0": 0.9, "depth_loss_1": 1.1}
        sum(c * m[k] for k, c in coef.items()).backward()
        res.append((m, d, i, lg))
    (mf, df, i_f, lf), (mt, dt, it, lt) = res
    for k in mt:
        assert abs(float(mf[k]) / float(mt[k]) - 1) < 2e-5, (k, float(mf[k]), float(mt[k]))
    for a, b in zip(df + [i_f, lf], dt + [it, lt]):
        assert rel_l2(host(a.grad), host(b.grad)) < 2e-5
    assert float(df[0].grad[~is_lidar].abs().max()) == 0.0  # camera rays get exactly zero


def test_lidar_quantile_threshold_matches_torch_quantile_on_ties_and_small_batches(ops):
    """the 0.95 quantile as torch computes it (fp32 rank, at::lerp): batches with many equal errors and n = 1"""
    from neurad_studio_amd.model_components.lidar_losses import LidarLossSettings, lidar_metrics

    cfg = LidarLossSettings()
    for n, vals in ((40, None), (1, None), (21, "ties")):
        torch.manual_seed(n)
        is_lidar = torch.ones(n, dtype=torch.bool, device="cuda")
        distance = torch.rand(n, 1, device="cuda") * 50 + 5
        depth = torch.rand(n, 1, device="cuda") * 60
        if vals == "ties":
            depth = distance + torch.tensor([0.0, 1.0, 2.0], device="cuda")[torch.arange(n, device="cuda") % 3][:, None]
        ret = torch.ones(n, dtype=torch.bool, device="cuda")
        out = {"depth": depth, "prop_depth_0": depth, "prop_depth_1": depth, "intensity": torch.rand(n, 1, device="cuda"),
               "ray_drop_logits": torch.randn(n, 1, device="cuda"), "non_nearby_weights_loss": torch.tensor(0.0, device="cuda"),
               "prop_weights_loss_0": torch.tensor(0.0, device="cuda"), "prop_weights_loss_1": torch.tensor(0.0, device="cuda")}
        tgt = torch.rand(n, 1, device="cuda")
        a = lidar_metrics(out, is_lidar, ret, distance, tgt, cfg, fused=True)
        b = lidar_metrics(out, is_lidar, ret, distance, tgt, cfg, fused=False)
        for k in ("depth_loss", "intensity_loss", "ray_drop_loss", "depth_loss_0"):
            fa, fb = float(a[k]), float(b[k])
            assert (np.isnan(fa) and np.isnan(fb)) or abs(fa - fb) <= 2e-5 * max(abs(fb), 1e-6), (n, vals, k, fa, fb)


def test_weighted_loss_sum_matches_python_sum(ops):
    from neurad_studio_amd.model_components.lidar_losses import WeightedLossSum

    terms = {k: torch.tensor(v, device="cuda", requires_grad=True) for k, v in (("a", 1.5), ("b", -2.0), ("c", 0.25))}
    mults = {"a": 0.1, "b": 3.0, "c": 7.0}
    total = WeightedLossSum(mults, "cuda")(terms)
    total.backward()
    assert abs(float(total) - sum(mults[k] * float(terms[k]) for k in terms)) < 1e-6
    for k in terms:
        assert abs(float(terms[k].grad) - mults[k]) < 1e-6


# ---- the composed step ------------------------------------------------------------------------------------------------
def _glue_model(g):
    from test_gpu_model_glue import build_model, bundle

    return build_model(g), bundle


def _losses(m, out, g, is_lidar, fused_metrics):
    from neurad_studio_amd.model_components.lidar_losses import LidarLossSettings, lidar_loss_dict, lidar_metrics, lidar_rows
    from neurad_studio_amd.model_components.losses import distortion_loss, zipnerf_interlevel_loss

    n_lidar = int(g["is_lidar"].sum())
    rows = lidar_rows(is_lidar, n_lidar)
    intensity, logits = m.decode_lidar(out["features"], rows=rows[0])
    outputs = dict(out, intensity=intensity, ray_drop_logits=logits)
    cfg = LidarLossSettings()
    did_return = dev(g["did_return"], torch.bool)[is_lidar]
    metrics = lidar_metrics(outputs, is_lidar, did_return, dev(g["directions_norm"])[is_lidar][:, None],
                            dev(g["lidar_points"])[:, 3:4], cfg, fused=fused_metrics, rows=rows)
    losses = lidar_loss_dict(metrics, cfg)
    lc = g["loss_cfg"]
    losses["interlevel_loss"] = float(lc[9]) * zipnerf_interlevel_loss(out["weights_list"], out["ray_samples_list"])
    losses["distortion_loss"] = float(lc[10]) * distortion_loss(out["weights_list"], out["ray_samples_list"])
    return metrics, losses, intensity, logits


def test_fused_training_step_vs_reference_golden_and_operator_path():
    """The fused training nodes produce the reference model's training outputs, lidar metrics, losses and gradients
    (model_train_glue.npz, 80 rays incl. lidar rays, appearance embedding, learnable beta) -- and agree with the
    operator-level path of the same model on every parameter gradient, element by element."""
    g = load_golden("model_train_glue")
    grads, outs = {}, {}
    for mode in ("fused", "operator"):
        m, bundle = _glue_model(g)
        m.train()
        m.sampler.eval(), m.field.eval()  # deterministic sampling, as in the generator
        for p in m.proposal_fields:
            p.eval()
        m.fused_training = mode == "fused"
        assert m.fused_training_possible() == (mode == "fused")
        out = m.get_nff_outputs(bundle(g), calc_lidar_losses=True)
        is_lidar = dev(g["is_lidar"], torch.bool)
        metrics, losses, intensity, logits = _losses(m, out, g, is_lidar, fused_metrics=mode == "fused")
        for k in ("features", "depth", "accumulation", "prop_depth_0", "prop_depth_1"):
            assert rel_l2(host(out[k]), g[k]) < TOL, (mode, k)
        for i in range(3):
            assert rel_l2(host(out["weights_list"][i][..., 0]), g[f"weights_{i}"]) < TOL, (mode, i)
        for i in range(2):
            assert abs(float(out[f"prop_weights_loss_{i}"]) / float(g[f"prop_weights_loss_{i}"]) - 1) < 1e-3
        assert rel_l2(host(intensity), g["intensity"]) < TOL and rel_l2(host(logits), g["ray_drop_logits"]) < TOL
        for k in ("depth_loss", "intensity_loss", "ray_drop_loss", "carving_loss", "depth_loss_0", "depth_loss_1",
                  "carving_loss_0", "carving_loss_1"):
            assert abs(float(metrics[k]) / float(g["metric_" + k]) - 1) < 2e-3, (mode, k, float(metrics[k]), float(g["metric_" + k]))
        for k, v in losses.items():
            assert abs(float(v) / float(g["loss_" + k]) - 1) < 2e-3, (mode, k)
        sum(losses.values()).backward()
        assert rel_l2(host(m.lidar_decoder.layers[0].weight.grad), g["g_lidar_decoder_w0"]) < 2e-3
        assert rel_l2(host(m.appearance_embedding.weight.grad), g["g_embedding"]) < 2e-3
        assert abs(float(m.field.sdf_to_density.beta.grad) / float(g["g_beta"]) - 1) < 5e-3
        assert m.proposal_fields[0].hashgrid.static_grid.hash_table.grad is None  # the late-binding quirk
        grads[mode] = {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}
        outs[mode] = out
    assert set(grads["fused"]) == set(grads["operator"])
    for n in grads["fused"]:
        assert rel_l2(host(grads["fused"][n]), host(grads["operator"][n])) < 5e-5, n
    # the reference's own table gradients, element by element where the golden carries them (sparse), else their mass
    for name, key in (("field.hashgrid.static_grid.hash_table", "g_field_table"),
                      ("proposal_fields.1.hashgrid.static_grid.hash_table", "g_prop1_table")):
        got = host(grads["fused"][name])
        if key + "_idx" in g:
            dense = np.zeros_like(got).reshape(-1)
            dense[g[key + "_idx"]] = g[key + "_val"]
            assert rel_l2(got.reshape(-1), dense) < 2e-3, name
        else:
            assert abs(float(np.abs(got).sum()) / float(g[key + "_abs_sum"]) - 1) < 5e-3


def test_fused_training_path_with_jitter_runs_and_matches_operator_path_given_the_same_draws():
    """training-mode jitter on: both paths draw torch.rand in the same order ([R,S+1] for the power bins, then one draw per
    PDF round), so with the same seed they walk the same samples"""
    from test_gpu_modules import bundle, small_model

    res = {}
    for mode in ("fused", "operator"):
        m = small_model(True).train()
        m.fused_training = mode == "fused"
        o, d, area, _ = synth.rays(96, 11)
        torch.manual_seed(1234)
        out = m.get_nff_outputs(bundle(o, d, area))
        assert len(out["weights_list"]) == 3 and out["weights_list"][0].shape == (96, 128, 1)
        loss = (out["features"].square().mean() + out["depth"].mean() * 1e-3 + out["accumulation"].mean()
                + sum(w.square().sum() for w in out["weights_list"]) + out["prop_depth_0"].mean() * 1e-3)
        loss.backward()
        res[mode] = (out, {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None})
    for k in ("features", "depth", "accumulation", "prop_depth_0", "prop_depth_1"):
        assert rel_l2(host(res["fused"][0][k]), host(res["operator"][0][k])) < 2e-5, k
    assert set(res["fused"][1]) == set(res["operator"][1])
    for n in res["fused"][1]:
        assert rel_l2(host(res["fused"][1][n]), host(res["operator"][1][n])) < 1e-4, n


def _actor_model():
    from neurad_studio_amd.model_components.dynamic_actors import DynamicActors, DynamicActorsConfig
    from neurad_studio_amd.models.neurad import NeuRADHotPath, NeuRADHotPathConfig
    from test_gpu_actors import trajectories

    torch.manual_seed(1)
    c = NeuRADHotPathConfig(appearance_dim=16)
    c.field.grid.static.log2_hashmap_size = 12
    c.field.grid.actor.log2_hashmap_size = 10
    c.field.sdf_beta = 3.0
    for pf in (c.sampling.proposal_field_1, c.sampling.proposal_field_2):
        pf.grid.static.log2_hashmap_size = 11
        pf.grid.actor.log2_hashmap_size = 9
    actors = DynamicActors(DynamicActorsConfig(), trajectories=trajectories())
    m = NeuRADHotPath(c, static_scale=100.0, num_sensors=2, duration=4.0, actors=actors).cuda().train()
    with torch.no_grad():
        m.field.hashgrid.static_grid.hash_table.mul_(500.0)
        for gr in m.field.hashgrid.actor_grids:
            gr.hash_table.mul_(3000.0)
        for p in m.proposal_fields:
            p.hashgrid.static_grid.hash_table.mul_(500.0)
            for gr in p.hashgrid.actor_grids:
                gr.hash_table.mul_(2000.0)
    m.sampler.eval()  # no sampling jitter; the actors' per-ray flip stays on (same draws in both paths under one seed)
    return m


def _actor_rays(R=384):
    from neurad_studio_amd.cameras.rays import RayBundle

    gen = torch.Generator().manual_seed(5)
    times = 1.0 + torch.rand(R, 1, generator=gen)  # all three trajectories exist in [1, 2]
    a = torch.arange(R) % 3  # look at actor a, where it is at the ray's time (test_gpu_actors.trajectories), from ~4 m
    tgt = torch.stack([12.0 + 2.0 * times[:, 0] + a, torch.tensor([8.0, -6.0, -5.0])[a], torch.full((R,), 0.5)], -1)
    side = torch.nn.functional.normalize(torch.randn(R, 3, generator=gen) * torch.tensor([1.0, 1.0, 0.15]), dim=-1)
    o = tgt + 4.0 * side
    d = torch.nn.functional.normalize(tgt + 0.3 * torch.randn(R, 3, generator=gen) - o, dim=-1)
    d[::4] = -d[::4]  # every fourth ray looks away
    return RayBundle(origins=o.cuda(), directions=d.cuda(), pixel_area=torch.full((R, 1), 2.7e-7, device="cuda"),
                     times=times.cuda(), metadata={"sensor_idxs": torch.randint(0, 2, (R, 1), generator=gen).cuda()})


def test_fused_training_with_dynamic_actors_matches_the_operator_level_path():
    """A scene with dynamic actors through the fused training nodes: samples inside a box take their encoding row and view
    direction from the differentiable actor branch (nrhip_field_fwd_train_ovr), the proposal rounds take the fields' own
    densities (static kernel + actor overlay).  Outputs and EVERY gradient -- static tables, actor grids, trajectory
    parameters, MLPs, beta, embedding -- against the operator-level path of the same model, which is pinned to the
    reference by the field_actors / field_actors_grads / proposal_actors goldens (tests/test_gpu_actors.py)."""
    from neurad_studio_amd.model_components.losses import distortion_loss, zipnerf_interlevel_loss

    res = {}
    for mode in ("fused", "operator"):
        m = _actor_model()
        m.fused_training = mode == "fused"
        assert m.field.hashgrid.has_actors() and m.fused_training_possible() == (mode == "fused")
        torch.manual_seed(77)
        out = m.get_nff_outputs(_actor_rays())
        loss = (out["features"].square().mean() + 1e-3 * out["depth"].mean() + out["accumulation"].mean()
                + 0.01 * zipnerf_interlevel_loss(out["weights_list"], out["ray_samples_list"])
                + 0.02 * distortion_loss(out["weights_list"], out["ray_samples_list"]) + 1e-3 * out["prop_depth_1"].mean())
        loss.backward()
        res[mode] = (out, {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None})
    for k in ("features", "depth", "accumulation", "prop_depth_0", "prop_depth_1"):
        assert rel_l2(host(res["fused"][0][k]), host(res["operator"][0][k])) < 5e-5, k
    gf, go = res["fused"][1], res["operator"][1]
    assert set(gf) == set(go), set(gf) ^ set(go)
    touched = [n for n in gf if ".actor_grids." in n and float(gf[n].abs().sum()) > 0]
    assert len(touched) >= 3, touched  # the scene exercises the actor grids of the field and of a proposal field
    assert any("actor_positions" in n for n in gf) and any("actor_rotations_6d" in n for n in gf)
    for n in gf:
        assert rel_l2(host(gf[n]), host(go[n])) < 2e-4, (n, rel_l2(host(gf[n]), host(go[n])))
