"""GPU tests of the host-side mirror (Field / Sampler / Renderer / hot-path model): the reference's plugin API
driving the HIP kernels, incl. the hand-written backward passes against the reference's autograd goldens."""
import numpy as np
import pytest
import torch

import neurad_oracle as O
import synth
from conftest import load_golden, rel_l2

pytestmark = pytest.mark.gpu
TOL = 1e-4


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    return t.detach().cpu().numpy()


def make_field(use_sdf, lg=11, num_multisamples=1):
    from neurad_studio_amd.fields.neurad_field import NeuRADField, NeuRADFieldConfig

    cfg = NeuRADFieldConfig(use_sdf=use_sdf, num_multisamples=num_multisamples)
    cfg.grid.static.log2_hashmap_size = lg
    f = NeuRADField(cfg, actors=None, static_scale=100.0).cuda()
    with torch.no_grad():
        f.hashgrid.static_grid.hash_table.copy_(dev(synth.hash_table(8 * 2**lg, 4, seed=51, scale=0.5)))
        for k, l in enumerate(f.mlp_geo.layers):
            w, b = synth.linear(l.out_features, l.in_features, 200 + 10 * k)
            l.weight.copy_(dev(w)), l.bias.copy_(dev(b))
        for k, l in enumerate(f.mlp_feature.layers):
            w, b = synth.linear(l.out_features, l.in_features, 300 + 10 * k)
            l.weight.copy_(dev(w)), l.bias.copy_(dev(b))
    return f


def bundle(o, d, area, fars=None):
    from neurad_studio_amd.cameras.rays import RayBundle

    R = o.shape[0]
    return RayBundle(origins=dev(o), directions=dev(d), pixel_area=dev(area)[:, None],
                     nears=torch.zeros(R, 1, device="cuda"),
                     fars=torch.full((R, 1), 20000.0, device="cuda") if fars is None else dev(fars)[:, None])


@pytest.mark.parametrize("fused_training", [True, False], ids=["fused-train", "operator-train"])
@pytest.mark.parametrize("tag", ["sdf", "density"])
def test_field_forward_backward_vs_reference_autograd(tag, fused_training):
    """Same parameters, inputs and upstream gradients as oracle/make_golden.py::golden_field; the golden gradients
    come from the reference's own autograd (B1).  Both training paths: the fused field kernel that saves its
    activations + hand-chained backward (FieldTrainFn), and the reference orchestration over operator autograd."""
    from neurad_studio_amd.field_components.field_heads import FieldHeadNames
    from neurad_studio_amd.model_components.ray_samplers import PowerSampler

    g = load_golden(f"field_{tag}")
    fld = make_field(tag == "sdf").eval()
    fld.fused_training = fused_training
    rb = bundle(g["o"], g["d"], g["area"])
    rs = PowerSampler(num_samples=12, lambda_=-1.0, scaling=0.1).eval()(rb)
    assert rel_l2(host(rs.frustums.starts[..., 0]), g["starts"]) < 1e-5
    key = FieldHeadNames.ALPHA if tag == "sdf" else FieldHeadNames.DENSITY
    with torch.no_grad():  # fused kernel
        out_f = fld(rs)
    out = fld(rs)  # training path (parameters require grad)
    assert rel_l2(host(out[FieldHeadNames.FEATURE]), g["feature"]) < TOL
    assert rel_l2(host(out_f[FieldHeadNames.FEATURE]), g["feature"]) < TOL
    assert rel_l2(host(out[key][..., 0]), g["alpha" if tag == "sdf" else "density"]) < TOL
    assert rel_l2(host(out_f[key][..., 0]), host(out[key][..., 0])) < 1e-5
    ((out[FieldHeadNames.FEATURE] * dev(g["g_feature"])).sum() + (out[key][..., 0] * dev(g["g_head"])).sum()).backward()
    tg = np.zeros((8 * 2**11, 4), np.float32)
    tg[g["tg_idx"]] = g["tg_val"]
    assert rel_l2(host(fld.hashgrid.static_grid.hash_table.grad), tg) < TOL
    for k, l in enumerate(fld.mlp_geo.layers):
        assert rel_l2(host(l.weight.grad), g[f"geo_dw{k}"]) < TOL and rel_l2(host(l.bias.grad), g[f"geo_db{k}"]) < TOL
    for k, l in enumerate(fld.mlp_feature.layers):
        assert rel_l2(host(l.weight.grad), g[f"feat_dw{k}"]) < TOL and rel_l2(host(l.bias.grad), g[f"feat_db{k}"]) < TOL
    if tag == "sdf":
        assert rel_l2(host(fld.sdf_to_density.beta.grad), g["dbeta"]) < TOL


def test_field_multisampled_vs_reference_golden():
    """NeuRADFieldConfig.num_multisamples = 3 (fields/neurad_field.py:67,134): forward (no-grad and training path) and the table /
    MLP gradients against the reference's own outputs for the same field and rays (oracle/make_golden_multisample.py)"""
    from neurad_studio_amd.field_components.field_heads import FieldHeadNames
    from neurad_studio_amd.model_components.ray_samplers import PowerSampler

    g, gm = load_golden("field_sdf"), load_golden("field_multisample")
    fld = make_field(True, num_multisamples=int(gm["num_multisamples"])).eval()
    assert not fld.fused_supported()  # M probes per frustum: not what the fused kernels compute
    rs = PowerSampler(num_samples=12, lambda_=-1.0, scaling=0.1).eval()(bundle(g["o"], g["d"], g["area"]))
    with torch.no_grad():
        out_e = fld(rs)
    out = fld(rs)
    for o_ in (out_e, out):
        assert rel_l2(host(o_[FieldHeadNames.FEATURE]), gm["feature"]) < TOL
        assert rel_l2(host(o_[FieldHeadNames.ALPHA][..., 0]), gm["alpha"]) < TOL
        assert rel_l2(host(o_[FieldHeadNames.SDF][..., 0]), gm["sdf"]) < TOL
    ((out[FieldHeadNames.FEATURE] * dev(g["g_feature"])).sum() + (out[FieldHeadNames.ALPHA][..., 0] * dev(g["g_head"])).sum()).backward()
    tg = np.zeros((8 * 2**11, 4), np.float32)
    tg[gm["tg_idx"]] = gm["tg_val"]
    assert rel_l2(host(fld.hashgrid.static_grid.hash_table.grad), tg) < TOL
    assert rel_l2(host(fld.mlp_geo.layers[0].weight.grad), gm["geo_dw0"]) < TOL
    # dynamic actors + multisampling is refused at construction
    from neurad_studio_amd.fields.neurad_field import NeuRADField, NeuRADFieldConfig
    from neurad_studio_amd.model_components.dynamic_actors import DynamicActors, DynamicActorsConfig
    from test_gpu_actors import trajectories

    with pytest.raises(NotImplementedError, match="num_multisamples"):
        NeuRADField(NeuRADFieldConfig(num_multisamples=2), actors=DynamicActors(DynamicActorsConfig(), trajectories=trajectories()),
                    static_scale=100.0)


def make_prop(seed, lg=11):
    from neurad_studio_amd.fields.neurad_field import NeuRADProposalField, NeuRADProposalFieldConfig

    c = NeuRADProposalFieldConfig()
    c.grid.static.log2_hashmap_size = lg
    p = NeuRADProposalField(c, actors=None, static_scale=100.0).cuda()
    w, _ = synth.linear(1, 6, seed + 1, bias=False)
    with torch.no_grad():
        p.hashgrid.static_grid.hash_table.copy_(dev(synth.hash_table(6 * 2**lg, 1, seed=seed, scale=2.0)))
        p.density_decoder.weight.copy_(dev(w + np.float32(0.3)))
    return p


def test_proposal_sampler_module_vs_reference_golden():
    """ProposalNetworkSampler with the reference's orchestration (density_fns callables) AND the fused kernel,
    against the chain the reference produced (incl. the late-binding quirk)."""
    from neurad_studio_amd.model_components.ray_samplers import PowerSampler, ProposalNetworkSampler

    g = load_golden("sampler_chain")
    props = [make_prop(91), make_prop(95)]
    sampler = ProposalNetworkSampler(num_proposal_samples_per_ray=(128, 64), num_nerf_samples_per_ray=32,
                                     num_proposal_network_iterations=2, single_jitter=True,
                                     initial_sampler=PowerSampler(lambda_=-1.0, scaling=0.1),
                                     update_sched=lambda x: 0).eval()
    density_fns = [lambda x: prop_field.get_density(x)[0] for prop_field in props]  # the reference's own idiom
    rb = bundle(g["o"], g["d"], g["area"], np.minimum(g["fars"], 20000.0))
    with torch.no_grad():
        rs, wl, rsl = sampler(rb, density_fns, pass_ray_samples=True)
        rs2, wl2, rsl2 = sampler.generate_fused(rb, [props[1], props[1]])
    for r, w in ((rs, wl), (rs2, wl2)):
        assert rel_l2(host(w[0][..., 0]), g["w0"]) < TOL and rel_l2(host(w[1][..., 0]), g["w1"]) < TOL
        assert rel_l2(host(r.frustums.starts[..., 0]), g["starts"]) < TOL
        assert rel_l2(host(r.frustums.ends[..., 0]), g["ends"]) < TOL
        assert rel_l2(host(r.spacing_ends[..., 0]), g["spe"]) < TOL


def small_model(use_sdf=True):
    from neurad_studio_amd.models.neurad import NeuRADHotPath, NeuRADHotPathConfig

    c = NeuRADHotPathConfig(appearance_dim=0)
    c.field.use_sdf = use_sdf
    c.field.sdf_beta = 3.0
    c.field.grid.static.log2_hashmap_size = 12
    c.sampling.proposal_field_1.grid.static.log2_hashmap_size = 11
    c.sampling.proposal_field_2.grid.static.log2_hashmap_size = 11
    torch.manual_seed(0)
    m = NeuRADHotPath(c, static_scale=100.0).cuda()
    with torch.no_grad():
        m.field.hashgrid.static_grid.hash_table.mul_(1000.0)  # O(1) features so alphas vary
        for p in m.proposal_fields:
            p.hashgrid.static_grid.hash_table.mul_(2000.0)
    return m


@pytest.mark.parametrize("use_sdf", [True, False])
def test_hot_path_model_eval_fused_vs_oracle_and_operator_path(use_sdf):
    m = small_model(use_sdf).eval()
    R = 48
    o, d, area, _ = synth.rays(R, 9)
    with torch.no_grad():
        out = m.get_nff_outputs(bundle(o, d, area / 9))  # _scale_pixel_area multiplies camera rays by 3^2
    # oracle chain with the same parameters
    h = lambda t: host(t)  # noqa: E731
    props = [O.ProposalParams(O.GridParams(h(p.hashgrid.static_grid.hash_table), 6, 128, 4096, 11), 100.0,
                              h(p.density_decoder.weight)) for p in m.proposal_fields]
    so = O.proposal_sampler(props, o, d, area, np.zeros(R), np.full(R, 20000.0, np.float32))
    f = m.field
    fp = O.FieldParams(O.GridParams(h(f.hashgrid.static_grid.hash_table), 8, 32, 8192, 12), 100.0,
                       [h(l.weight) for l in f.mlp_geo.layers], [h(l.bias) for l in f.mlp_geo.layers],
                       [h(l.weight) for l in f.mlp_feature.layers], [h(l.bias) for l in f.mlp_feature.layers],
                       beta=3.0, use_sdf=use_sdf)
    ref = O.render_rays(fp, o, d, area, so.starts, so.ends)
    assert rel_l2(host(out["features"]), ref["features"]) < TOL
    assert rel_l2(host(out["accumulation"]), ref["accumulation"]) < TOL
    assert rel_l2(host(out["depth"]), ref["depth"]) < TOL
    # operator-level (training) path in eval mode (no jitter) must agree with the fused kernels
    out2 = m.get_nff_outputs(bundle(o, d, area / 9))
    for k in ("features", "accumulation", "depth", "prop_depth_0", "prop_depth_1"):
        assert rel_l2(host(out2[k]), host(out[k])) < 2e-5, k
    # gradients reach every trainable tensor of the path that the reference trains
    (out2["features"].sum() + out2["depth"].sum()).backward()
    assert m.field.hashgrid.static_grid.hash_table.grad.abs().sum() > 0
    assert all(l.weight.grad is not None for l in m.field.mlp_geo.layers)


def test_training_mode_runs_with_jitter_and_proposal_grads():
    m = small_model(True).train()
    R = 64
    o, d, area, _ = synth.rays(R, 11)
    out = m.get_nff_outputs(bundle(o, d, area))
    assert len(out["weights_list"]) == 3 and out["weights_list"][0].shape == (R, 128, 1)
    loss = out["features"].square().mean() + sum(w.square().sum() for w in out["weights_list"][:2])
    loss.backward()
    # the late-binding quirk: only proposal_fields[1] is ever evaluated -> [0] gets no gradient
    assert m.proposal_fields[0].hashgrid.static_grid.hash_table.grad is None
    assert m.proposal_fields[1].hashgrid.static_grid.hash_table.grad.abs().sum() > 0
    assert m.proposal_fields[1].density_decoder.weight.grad.abs().sum() > 0


def test_nerfacc_shaped_shim_and_renderers():
    from neurad_studio_amd.model_components.renderers import AccumulationRenderer, FeatureRenderer
    from neurad_studio_amd.shims import nerfacc

    a = dev(synth.uniform((9, 40), 0, 0.3, 1)).requires_grad_(True)
    w, t = nerfacc.render_weight_from_alpha(a)
    feats = dev(synth.normal((9, 40, 32), 2))
    out = FeatureRenderer()(features=feats, weights=w[..., None])
    acc = AccumulationRenderer()(weights=w[..., None])
    (out.sum() + acc.sum()).backward()
    rw, _ = O.render_weight_from_alpha(host(a))
    assert rel_l2(host(out), O.accumulate_along_rays(rw, host(feats))) < 2e-5
    assert a.grad is not None and torch.isfinite(a.grad).all()


def test_tinycudann_shaped_shim_matches_torch_path_oracle():
    """tcnn.Encoding / Network / NetworkWithInputEncoding call forms of encodings.py:370-373, mlp.py:109-113,251-268."""
    from neurad_studio_amd.shims import tinycudann as tcnn

    growth = float(np.exp((np.log(1024) - np.log(16)) / 15))
    enc_cfg = {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": 12,
               "base_resolution": 16, "per_level_scale": growth}
    net_cfg = {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None", "n_neurons": 64,
               "n_hidden_layers": 1}
    m = tcnn.NetworkWithInputEncoding(3, 16, enc_cfg, net_cfg).cuda()
    x = synth.uniform((777, 3), 0, 1, 4)
    y = m(dev(x))
    table = host(m.encoding.params).reshape(-1, 2)
    sc = O.hash_scalings(16, 16, 1024)
    np.testing.assert_array_equal(m.encoding.spec.scalings.numpy(), sc)
    ref = O.mlp_fwd(O.hashgrid_fwd(x, table, sc, 2**12), [host(l.weight) for l in m.network.layers],
                    [host(l.bias) for l in m.network.layers])
    assert rel_l2(host(y), ref) < TOL
    y.square().sum().backward()
    assert m.encoding.params.grad.abs().sum() > 0 and m.network.layers[0].weight.grad.abs().sum() > 0
    sh = tcnn.Encoding(3, {"otype": "SphericalHarmonics", "degree": 4}).cuda()
    d = synth.uniform((50, 3), 0, 1, 5)
    assert rel_l2(host(sh(dev(d))), O.sh_deg4(d)) < 1e-6


def test_full_size_neurad_default_chain_properties():
    """BASELINE config[2]/[3] shape at full size: default NeuRAD grids (static 8 x 2^22 x 4 = 537 MB, proposals
    6 x 2^20), 8192 rays, sampler (128, 64) -> 32 + field + compositing.  Size-independent properties: fused eval path
    == operator-level path, bins sorted inside [0, sky], accumulation in [0, 1], finite, ray-permutation covariance."""
    from neurad_studio_amd.models.neurad import NeuRADHotPath, NeuRADHotPathConfig

    torch.manual_seed(0)
    m = NeuRADHotPath(NeuRADHotPathConfig(appearance_dim=0), static_scale=100.0).cuda().eval()
    with torch.no_grad():
        m.field.hashgrid.static_grid.hash_table.mul_(300.0)
        for p in m.proposal_fields:
            p.hashgrid.static_grid.hash_table.mul_(1000.0)
    R = 8192
    o, d, area, _ = synth.rays(R, 21)
    with torch.no_grad():
        out = m.get_nff_outputs(bundle(o, d, area / 9))
        rs, wl, rsl = m.sampler.generate_fused(bundle(o, d, area), [m.proposal_fields[1]] * 2)
    assert all(torch.isfinite(v).all() for v in out.values())
    acc = out["accumulation"]
    assert float(acc.min()) >= -1e-6 and float(acc.max()) <= 1 + 1e-5
    e = torch.cat([rs.frustums.starts[..., 0], rs.frustums.ends[:, -1:, 0]], -1)
    assert bool((e[:, 1:] >= e[:, :-1]).all()) and float(e.min()) >= 0 and float(e.max()) <= 20000.0 + 1e-2
    assert wl[0].shape == (R, 128, 1) and wl[1].shape == (R, 64, 1)
    assert float(wl[0].sum(-2).max()) <= 1 + 1e-4  # proposal weights are a sub-probability along each ray
    perm = torch.randperm(R, device="cuda").cpu().numpy()
    with torch.no_grad():
        outp = m.get_nff_outputs(bundle(o[perm], d[perm], area[perm] / 9))
    assert torch.equal(outp["features"], out["features"][torch.from_numpy(perm).cuda()])
    # operator-level path on a slice (grad-enabled call in eval mode -> no jitter) agrees with the fused kernels
    sl = slice(0, 512)
    out2 = m.get_nff_outputs(bundle(o[sl], d[sl], area[sl] / 9))
    for k in ("features", "accumulation", "depth"):
        assert rel_l2(host(out2[k]), host(out[k][sl])) < 5e-5, k


def test_full_size_neurad_default_chain_vs_c_oracle():
    """BASELINE config[2]/[3] at FULL size (static 8 x 2^22 x 4, proposals 6 x 2^20, 8192 rays, (128, 64) -> 32) against
    the independent C restatement of the whole chain (oracle/neurad_oracle_c.c: S1-S5 + M1 + H1-H4 + F1-F4 + C1/C2,
    pinned to the numpy oracle -- and through it to the reference's goldens -- by tests/test_oracle_c.py)."""
    import oracle_c
    from neurad_studio_amd.models.neurad import NeuRADHotPath, NeuRADHotPathConfig

    torch.manual_seed(0)
    m = NeuRADHotPath(NeuRADHotPathConfig(appearance_dim=0), static_scale=100.0).cuda().eval()
    with torch.no_grad():
        m.field.hashgrid.static_grid.hash_table.mul_(300.0)
        for p in m.proposal_fields:
            p.hashgrid.static_grid.hash_table.mul_(1000.0)
    R = 8192
    o, d, area, _ = synth.rays(R, 21)
    with torch.no_grad():
        out = m.get_nff_outputs(bundle(o, d, area / 9))
        rs, wl, _ = m.sampler.generate_fused(bundle(o, d, area), [m.proposal_fields[1]] * 2)
    props = [O.ProposalParams(O.GridParams(host(p.hashgrid.static_grid.hash_table), 6, 128, 4096, 20), 100.0,
                              host(p.density_decoder.weight)) for p in m.proposal_fields]
    so = oracle_c.proposal_sampler(props, o, d, area, np.zeros(R, np.float32), np.full(R, 20000.0, np.float32))
    for i in range(2):
        assert rel_l2(host(wl[i][..., 0]), so["prop_weights"][i]) < TOL, i
    # final bins: inverse-CDF sampling is continuous but ill-conditioned where the CDF is flat -> robust comparison
    for k, got in (("starts", host(rs.frustums.starts[..., 0])), ("ends", host(rs.frustums.ends[..., 0]))):
        got, want = got[:, :-1], so[k][:, :-1]  # (the last end is the sky stretch, applied by the model, not the sampler)
        bad = np.abs(got - want) > 2e-4 * np.abs(want) + 2e-5
        assert bad.mean() < 1e-3, (k, bad.mean())
    f = m.field
    fp = O.FieldParams(O.GridParams(host(f.hashgrid.static_grid.hash_table), 8, 32, 8192, 22), 100.0,
                       [host(l.weight) for l in f.mlp_geo.layers], [host(l.bias) for l in f.mlp_geo.layers],
                       [host(l.weight) for l in f.mlp_feature.layers], [host(l.bias) for l in f.mlp_feature.layers],
                       beta=float(f.sdf_to_density.beta), use_sdf=True)
    ref = oracle_c.render_fwd(fp, o, d, area, so["starts"], so["ends"])
    for k in ("features", "accumulation", "depth"):
        assert rel_l2(host(out[k]), ref[k]) < TOL, k


def test_neurader_sized_grids_take_the_fused_kernels_and_match_the_c_oracle():
    """The reference's larger methods (`neurader`, `neuradest`, the `-paper` / `-scaleopt` variants,
    configs/method_configs.py:468-510) change the grids only: base_res and max_res x 2, log2_hashmap_size + 1 for the field
    AND both proposal fields (static 8 x 2^23 x 4 = 1.07 GB).  Same kernel instantiations as the default method -- this pins
    that shape against the C restatement of the chain on 2048 rays."""
    import oracle_c
    from neurad_studio_amd.models.neurad import NeuRADHotPath, NeuRADHotPathConfig

    torch.manual_seed(0)
    cfg = NeuRADHotPathConfig(appearance_dim=0)
    for fc in (cfg.field, cfg.sampling.proposal_field_1, cfg.sampling.proposal_field_2):
        fc.grid.static.max_res *= 2
        fc.grid.static.base_res *= 2
        fc.grid.static.log2_hashmap_size += 1
    m = NeuRADHotPath(cfg, static_scale=100.0).cuda().eval()
    with torch.no_grad():
        m.field.hashgrid.static_grid.hash_table.mul_(300.0)
        for p in m.proposal_fields:
            p.hashgrid.static_grid.hash_table.mul_(1000.0)
    R = 2048
    o, d, area, _ = synth.rays(R, 22)
    with torch.no_grad():
        assert m.fused_eval_possible()  # the two fused kernels, not the operator-level path
        out = m.get_nff_outputs(bundle(o, d, area / 9))
        rs, wl, _ = m.sampler.generate_fused(bundle(o, d, area), [m.proposal_fields[1]] * 2)
    pg, fg = cfg.sampling.proposal_field_1.grid.static, cfg.field.grid.static
    props = [O.ProposalParams(O.GridParams(host(p.hashgrid.static_grid.hash_table), pg.num_levels, pg.base_res, pg.max_res,
                                           pg.log2_hashmap_size), 100.0, host(p.density_decoder.weight))
             for p in m.proposal_fields]
    so = oracle_c.proposal_sampler(props, o, d, area, np.zeros(R, np.float32), np.full(R, 20000.0, np.float32))
    for i in range(2):
        assert rel_l2(host(wl[i][..., 0]), so["prop_weights"][i]) < TOL, i
    f = m.field
    fp = O.FieldParams(O.GridParams(host(f.hashgrid.static_grid.hash_table), fg.num_levels, fg.base_res, fg.max_res,
                                    fg.log2_hashmap_size), 100.0,
                       [host(l.weight) for l in f.mlp_geo.layers], [host(l.bias) for l in f.mlp_geo.layers],
                       [host(l.weight) for l in f.mlp_feature.layers], [host(l.bias) for l in f.mlp_feature.layers],
                       beta=float(f.sdf_to_density.beta), use_sdf=True)
    ref = oracle_c.render_fwd(fp, o, d, area, so["starts"], so["ends"])
    for k in ("features", "accumulation", "depth"):
        assert rel_l2(host(out[k]), ref[k]) < TOL, k
