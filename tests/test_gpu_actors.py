"""GPU parity of the dynamic-actor path against the reference's own outputs
(tests/golden/field_actors.npz) and the oracle."""
import numpy as np
import pytest
import torch

import neurad_oracle as O
import synth
from conftest import load_golden, rel_l2
from test_oracle_actors import actor_params, field_params

pytestmark = pytest.mark.gpu
TOL = 1e-4


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    return t.detach().cpu().numpy()


def trajectories():
    """same synthetic trajectories as oracle/make_golden_actors.py"""
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


def make_field():
    from neurad_studio_amd.fields.neurad_field import NeuRADField, NeuRADFieldConfig
    from neurad_studio_amd.model_components.dynamic_actors import DynamicActors, DynamicActorsConfig

    actors = DynamicActors(DynamicActorsConfig(), trajectories=trajectories())
    cfg = NeuRADFieldConfig()
    cfg.grid.static.log2_hashmap_size = 11
    cfg.grid.actor.log2_hashmap_size = 9
    fld = NeuRADField(cfg, actors=actors, static_scale=100.0).cuda().eval()
    with torch.no_grad():
        fld.hashgrid.static_grid.hash_table.copy_(dev(synth.hash_table(8 * 2**11, 4, seed=51, scale=0.5)))
        for i, g in enumerate(fld.hashgrid.actor_grids):
            g.hash_table.copy_(dev(synth.hash_table(4 * 2**9, 4, seed=400 + i, scale=0.7)))
        for k, l in enumerate(fld.mlp_geo.layers):
            w, b = synth.linear(l.out_features, l.in_features, 200 + 10 * k)
            l.weight.copy_(dev(w)), l.bias.copy_(dev(b))
        for k, l in enumerate(fld.mlp_feature.layers):
            w, b = synth.linear(l.out_features, l.in_features, 300 + 10 * k)
            l.weight.copy_(dev(w)), l.bias.copy_(dev(b))
    return fld


def test_actor_state_matches_reference_buffers():
    g = load_golden("field_actors")
    fld = make_field()
    act = fld.hashgrid.actors
    np.testing.assert_array_equal(host(act.actor_present_at_time), g["present"])
    assert rel_l2(host(act.actor_positions), g["positions"]) < 1e-7
    assert rel_l2(host(act.actor_rotations_6d), g["rotations_6d"]) < 1e-7
    assert rel_l2(host(act.actor_bounds()), g["sizes"] / 2 + g["padding"]) < 1e-7


def test_field_with_actors_vs_reference_golden():
    from neurad_studio_amd.cameras.rays import RayBundle
    from neurad_studio_amd.field_components.field_heads import FieldHeadNames

    g = load_golden("field_actors")
    fld = make_field()
    R = g["o"].shape[0]
    rb = RayBundle(origins=dev(g["o"]), directions=dev(g["d"]), pixel_area=dev(g["area"])[:, None],
                   times=dev(g["times"])[:, None], nears=torch.zeros(R, 1, device="cuda"),
                   fars=torch.full((R, 1), 60.0, device="cuda"))
    rs = rb.get_ray_samples(dev(g["starts"])[..., None], dev(g["ends"])[..., None])
    # hit set == the reference's (ray, sample) pairs
    o, d, a = rs.frustums.per_ray()
    spec, cand = fld.hashgrid.prepare_actors(o, d, a, dev(g["starts"]), dev(g["ends"]), dev(g["times"]))
    feats = torch.zeros((R * g["starts"].shape[1], 32), device="cuda")
    from neurad_studio_amd import ops
    dirs, hit = ops.actor_encode(spec, cand, o, d, a, dev(g["starts"]), dev(g["ends"]), feats)
    want = np.zeros(g["starts"].shape, bool)
    want[g["hit_ray"], g["hit_sample"]] = True
    np.testing.assert_array_equal(host(hit).reshape(want.shape) >= 0, want)
    assert rel_l2(host(dirs).reshape(g["directions"].shape), g["directions"]) < 1e-6
    with torch.no_grad():
        out = fld(rs)
    assert rel_l2(host(out[FieldHeadNames.FEATURE]), g["feature"]) < TOL
    assert rel_l2(host(out[FieldHeadNames.ALPHA][..., 0]), g["alpha"]) < TOL
    # encoding (static + overwritten actor rows) through the module API
    enc, dd = fld.hashgrid(rs)
    assert rel_l2(host(enc), g["enc"]) < TOL
    # training-style call: static-table gradient exists and is zero for rows that actors overwrote
    fld.hashgrid.config.actor.flip_prob = 0.0
    out2 = fld(rs)
    out2[FieldHeadNames.FEATURE].sum().backward()
    assert fld.hashgrid.static_grid.hash_table.grad.abs().sum() > 0


def test_actor_training_flip_injected():
    """training-mode per-ray x flip (neurad_encoding.py:212-219) with injected +-1 per ray, vs the oracle"""
    from neurad_studio_amd import ops

    g = load_golden("field_actors")
    fld = make_field()
    R, S = g["starts"].shape
    o, d, a = dev(g["o"]), dev(g["d"]), dev(g["area"])
    spec, cand = fld.hashgrid.prepare_actors(o, d, a, dev(g["starts"]), dev(g["ends"]), dev(g["times"]))
    flip = np.where(np.arange(R) % 3 == 0, -1.0, 1.0).astype(np.float32)
    grid = O.GridParams(synth.hash_table(8 * 2**11, 4, seed=51, scale=0.5), 8, 32, 8192, 11)
    ref, rdirs = O.encode_with_actors(grid, 100.0, actor_params(g), g["o"], g["d"], g["area"], g["starts"], g["ends"],
                                      g["times"], ray_flip=flip)
    feats = ops.encode_fwd(fld.hashgrid.static_grid.spec, fld.hashgrid.static_grid.hash_table.detach(), 100.0, o, d, a,
                           dev(g["starts"]), dev(g["ends"]))
    dirs, hit = ops.actor_encode(spec, cand, o, d, a, dev(g["starts"]), dev(g["ends"]), feats, dev(flip))
    assert rel_l2(host(feats), ref) < TOL
    assert rel_l2(host(dirs).reshape(R, S, 3), rdirs) < 1e-6


def test_proposal_density_with_actors_vs_oracle():
    from neurad_studio_amd.cameras.rays import RayBundle
    from neurad_studio_amd.fields.neurad_field import NeuRADProposalField, NeuRADProposalFieldConfig
    from neurad_studio_amd.model_components.dynamic_actors import DynamicActors, DynamicActorsConfig

    g = load_golden("field_actors")
    actors = DynamicActors(DynamicActorsConfig(), trajectories=trajectories())
    c = NeuRADProposalFieldConfig()
    c.grid.static.log2_hashmap_size = 11
    c.grid.actor.log2_hashmap_size = 8
    p = NeuRADProposalField(c, actors=actors, static_scale=100.0).cuda().eval()
    w, _ = synth.linear(1, 6, 77, bias=False)
    with torch.no_grad():
        p.hashgrid.static_grid.hash_table.copy_(dev(synth.hash_table(6 * 2**11, 1, seed=91, scale=2.0)))
        for i, gr in enumerate(p.hashgrid.actor_grids):
            gr.hash_table.copy_(dev(synth.hash_table(4 * 2**8, 1, seed=500 + i, scale=1.5)))
        p.density_decoder.weight.copy_(dev(w + np.float32(0.3)))
    R = g["o"].shape[0]
    rb = RayBundle(origins=dev(g["o"]), directions=dev(g["d"]), pixel_area=dev(g["area"])[:, None],
                   times=dev(g["times"])[:, None], nears=torch.zeros(R, 1, device="cuda"),
                   fars=torch.full((R, 1), 60.0, device="cuda"))
    rs = rb.get_ray_samples(dev(g["starts"])[..., None], dev(g["ends"])[..., None])
    with torch.no_grad():
        dens = host(p.get_density(rs)[0][..., 0])
    # oracle: static proposal density, then exp(decoder . padded actor feats) on the hit samples
    grid = O.GridParams(synth.hash_table(6 * 2**11, 1, seed=91, scale=2.0), 6, 128, 4096, 11)
    pp = O.ProposalParams(grid, 100.0, w + np.float32(0.3))
    ap = actor_params(g)
    ap.grids = [O.GridParams(synth.hash_table(4 * 2**8, 1, seed=500 + i, scale=1.5), 4, 64, 1024, 8) for i in range(3)]
    enc, _ = O.encode_with_actors(grid, 100.0, ap, g["o"], g["d"], g["area"], g["starts"], g["ends"], g["times"])
    ref = np.exp(enc @ pp.decoder_w.T).reshape(g["starts"].shape)
    assert rel_l2(dens, ref) < TOL


def test_hashgrid_dx_vs_reference_autograd():
    from neurad_studio_amd import ops

    g = load_golden("field_actors_grads")
    spec = ops.GridSpec(4, 4, 9, 64, 1024)
    table = dev(synth.hash_table(4 * 2**9, 4, seed=400, scale=0.7))
    gx = ops.hashgrid_bwd_input(spec, table, dev(g["hx"]), dev(g["hgy"]))
    assert rel_l2(host(gx), g["hdx"]) < TOL


def test_actor_gradients_vs_reference_autograd():
    """B1 with actors: static table, per-actor grids and the trajectory parameters get the reference's gradients."""
    from neurad_studio_amd.cameras.rays import RayBundle
    from neurad_studio_amd.field_components.field_heads import FieldHeadNames

    g, gg = load_golden("field_actors"), load_golden("field_actors_grads")
    fld = make_field()
    R = g["o"].shape[0]
    rb = RayBundle(origins=dev(g["o"]), directions=dev(g["d"]), pixel_area=dev(g["area"])[:, None],
                   times=dev(g["times"])[:, None], nears=torch.zeros(R, 1, device="cuda"),
                   fars=torch.full((R, 1), 60.0, device="cuda"))
    rs = rb.get_ray_samples(dev(g["starts"])[..., None], dev(g["ends"])[..., None])
    out = fld(rs)
    assert rel_l2(host(out[FieldHeadNames.FEATURE]), g["feature"]) < TOL
    ((out[FieldHeadNames.FEATURE] * dev(gg["g_feature"])).sum()
     + (out[FieldHeadNames.ALPHA][..., 0] * dev(gg["g_alpha"])).sum()).backward()
    tg = np.zeros((8 * 2**11, 4), np.float32)
    tg[gg["tg_idx"]] = gg["tg_val"]
    assert rel_l2(host(fld.hashgrid.static_grid.hash_table.grad), tg) < TOL
    for i, gr in enumerate(fld.hashgrid.actor_grids):
        assert rel_l2(host(gr.hash_table.grad), gg[f"ag{i}"]) < TOL, i
    act = fld.hashgrid.actors
    assert rel_l2(host(act.actor_positions.grad), gg["dpos"]) < 1e-3
    assert rel_l2(host(act.actor_rotations_6d.grad), gg["drot"]) < 1e-3


def test_many_actors_along_a_ray_no_candidate_cap():
    """20 parked cars in a row: rays along the row cross the bounding spheres of up to 20 actors (round 1 raised above
    8; the reference has no limit, neurad_encoding.py:225-263).  Hit set, features and alphas against the oracle, and
    the differentiable path still hands every actor grid its gradient."""
    from neurad_studio_amd.cameras.rays import RayBundle
    from neurad_studio_amd.field_components.field_heads import FieldHeadNames
    from neurad_studio_amd.fields.neurad_field import NeuRADField, NeuRADFieldConfig
    from neurad_studio_amd.model_components.dynamic_actors import DynamicActors, DynamicActorsConfig

    A = 20
    ts = torch.tensor([0.0, 1.0])
    trajs = []
    for a in range(A):
        p = torch.eye(4).repeat(2, 1, 1)
        p[:, :3, 3] = torch.tensor([6.0 + 5.0 * a, 0.3 * (a % 3 - 1), 0.4])
        trajs.append({"timestamps": ts.clone(), "poses": p, "dims": torch.tensor([2.0, 4.6, 1.6]),
                      "symmetric": torch.tensor(True), "deformable": torch.tensor(False)})
    actors = DynamicActors(DynamicActorsConfig(), trajectories=trajs)
    cfg = NeuRADFieldConfig()
    cfg.grid.static.log2_hashmap_size = 11
    cfg.grid.actor.log2_hashmap_size = 9
    fld = NeuRADField(cfg, actors=actors, static_scale=100.0).cuda().eval()
    fp = field_params()
    with torch.no_grad():
        fld.hashgrid.static_grid.hash_table.copy_(dev(fp.grid.table))
        tabs = [synth.hash_table(4 * 2**9, 4, seed=900 + i, scale=0.7) for i in range(A)]
        for gr, t in zip(fld.hashgrid.actor_grids, tabs):
            gr.hash_table.copy_(dev(t))
        for layers, ws, bs in ((fld.mlp_geo.layers, fp.geo_w, fp.geo_b), (fld.mlp_feature.layers, fp.feat_w, fp.feat_b)):
            for l, w, b in zip(layers, ws, bs):
                l.weight.copy_(dev(w)), l.bias.copy_(dev(b))
    R, S = 32, 64
    o = (synth.normal((R, 3), 3) * np.array([0.5, 0.3, 0.1], np.float32)).astype(np.float32)
    tgt = np.stack([np.full(R, 110.0), synth.uniform((R,), -0.6, 0.6, 4), synth.uniform((R,), 0.2, 0.6, 5)], -1)
    d = (tgt - o).astype(np.float32)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    times = synth.uniform((R,), 0.0, 1.0, 6)
    area = np.full((R,), 2.43e-6, np.float32)
    edges = np.linspace(0.0, 108.0, S + 1, dtype=np.float32)[None].repeat(R, 0)
    st, en = np.ascontiguousarray(edges[:, :-1]), np.ascontiguousarray(edges[:, 1:])
    ap = O.ActorParams(host(actors.unique_timestamps), host(actors.actor_positions), host(actors.actor_rotations_6d),
                       host(actors.actor_present_at_time), host(actors.actor_sizes), host(actors.actor_padding),
                       [O.GridParams(t, 4, 64, 1024, 9) for t in tabs], actor_scale=10.0)
    ref = O.field_fwd_actors(fp, ap, o, d, area, st, en, times)
    rb = RayBundle(origins=dev(o), directions=dev(d), pixel_area=dev(area)[:, None], times=dev(times)[:, None])
    rs = rb.get_ray_samples(dev(st)[..., None], dev(en)[..., None])
    spec, cand = fld.hashgrid.prepare_actors(dev(o), dev(d), dev(area), dev(st), dev(en), dev(times))
    assert cand[1].shape == (R, A) and int(cand[0].max()) > 8, "the scene must exceed the old per-ray cap"
    with torch.no_grad():
        out = fld(rs)
    assert rel_l2(host(out[FieldHeadNames.FEATURE]), ref["feature"]) < TOL
    assert rel_l2(host(out[FieldHeadNames.ALPHA][..., 0]), ref["alpha"]) < TOL
    out2 = fld(rs)
    (out2[FieldHeadNames.FEATURE].square().sum() + out2[FieldHeadNames.ALPHA].sum()).backward()
    touched = sum(int(gr.hash_table.grad is not None and float(gr.hash_table.grad.abs().sum()) > 0)
                  for gr in fld.hashgrid.actor_grids)
    assert touched >= 12, touched
    assert actors.actor_positions.grad is not None and float(actors.actor_positions.grad.abs().sum()) > 0


def _composite_reference(feature, alpha, starts, ends):
    """C1 + C2 on per-sample outputs (oracle: nerfacc's exclusive cumprod + the sky residual of models/neurad.py:381)"""
    w, _ = O.render_weight_from_alpha(alpha)
    return O.composite(w, feature, starts, ends)


def test_fused_render_with_actors_vs_reference_golden():
    """nrhip_render_fwd_actors (device-side ray split + static kernel + per-sample table select kernel) against the
    REFERENCE's per-sample field outputs for the actor scene (golden field_actors: feature, alpha), composited."""
    g = load_golden("field_actors")
    fld = make_field()
    starts, ends = g["starts"], g["ends"]
    with torch.no_grad():
        feats, depth, acc, w = fld.render(dev(g["o"]), dev(g["d"]), dev(g["area"]), dev(starts), dev(ends), return_weights=True,
                                          times=dev(g["times"]))
    want_f, want_d, want_a = _composite_reference(g["feature"], g["alpha"], starts, ends)
    assert want_f.shape == feats.shape
    assert rel_l2(host(feats), want_f) < TOL
    assert rel_l2(host(acc), want_a) < TOL
    assert rel_l2(host(depth), want_d) < TOL
    assert rel_l2(host(w), O.render_weight_from_alpha(g["alpha"])[0]) < TOL
    # the scene really exercises both kernels: some rays have candidates, some do not
    cnt = fld.hashgrid.prepare_actors(dev(g["o"]), dev(g["d"]), dev(g["area"]), dev(starts), dev(ends), dev(g["times"]))[1][0]
    assert 0 < int((cnt > 0).sum()) and len(g["hit_ray"]) > 0


@pytest.mark.parametrize("with_order", [False, True])
def test_fused_render_many_actors_vs_oracle_and_operator_path(with_order):
    """20 actors in a row (up to 20 candidates per ray, samples of one 16-sample tile inside different boxes), mixed with
    rays that pass no actor: fused == oracle == operator-level path, with and without a locality order."""
    from neurad_studio_amd import ops
    from neurad_studio_amd.fields.neurad_field import NeuRADField, NeuRADFieldConfig
    from neurad_studio_amd.model_components.dynamic_actors import DynamicActors, DynamicActorsConfig

    A = 20
    ts = torch.tensor([0.0, 1.0])
    trajs = []
    for a in range(A):
        p = torch.eye(4).repeat(2, 1, 1)
        p[:, :3, 3] = torch.tensor([6.0 + 5.0 * a, 0.3 * (a % 3 - 1), 0.4])
        trajs.append({"timestamps": ts.clone(), "poses": p, "dims": torch.tensor([2.0, 4.6, 1.6]),
                      "symmetric": torch.tensor(True), "deformable": torch.tensor(False)})
    actors = DynamicActors(DynamicActorsConfig(), trajectories=trajs)
    cfg = NeuRADFieldConfig()
    cfg.grid.static.log2_hashmap_size = 11
    cfg.grid.actor.log2_hashmap_size = 9
    fld = NeuRADField(cfg, actors=actors, static_scale=100.0).cuda().eval()
    fp = field_params()
    with torch.no_grad():
        fld.hashgrid.static_grid.hash_table.copy_(dev(fp.grid.table))
        tabs = [synth.hash_table(4 * 2**9, 4, seed=900 + i, scale=0.7) for i in range(A)]
        for gr, t in zip(fld.hashgrid.actor_grids, tabs):
            gr.hash_table.copy_(dev(t))
        for layers, ws, bs in ((fld.mlp_geo.layers, fp.geo_w, fp.geo_b), (fld.mlp_feature.layers, fp.feat_w, fp.feat_b)):
            for l, w, b in zip(layers, ws, bs):
                l.weight.copy_(dev(w)), l.bias.copy_(dev(b))
        fld.sdf_to_density.beta.fill_(fp.beta)
    R, S = 301, 72  # ragged: not a multiple of the 4 rays per workgroup, last tile half full
    o = (synth.normal((R, 3), 3) * np.array([0.5, 0.3, 0.1], np.float32)).astype(np.float32)
    tgt = np.stack([np.full(R, 110.0), synth.uniform((R,), -0.6, 0.6, 4), synth.uniform((R,), 0.2, 0.6, 5)], -1)
    tgt[::3] = np.stack([synth.uniform((R,), -50, 50, 7), np.full(R, 90.0), synth.uniform((R,), 5, 40, 8)], -1)[::3]  # away
    d = (tgt - o).astype(np.float32)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    times = synth.uniform((R,), 0.0, 1.0, 6)
    area = np.full((R,), 2.43e-6, np.float32)
    edges = np.linspace(0.0, 108.0, S + 1, dtype=np.float32)[None].repeat(R, 0)
    st, en = np.ascontiguousarray(edges[:, :-1]), np.ascontiguousarray(edges[:, 1:])
    ap = O.ActorParams(host(actors.unique_timestamps), host(actors.actor_positions), host(actors.actor_rotations_6d),
                       host(actors.actor_present_at_time), host(actors.actor_sizes), host(actors.actor_padding),
                       [O.GridParams(t, 4, 64, 1024, 9) for t in tabs], actor_scale=10.0)
    ref = O.field_fwd_actors(fp, ap, o, d, area, st, en, times)
    want_f, want_d, want_a = _composite_reference(ref["feature"], ref["alpha"], st, en)
    order = ops.ray_order(dev(o), dev(d), 100.0) if with_order else None
    with torch.no_grad():
        feats, depth, acc = fld.render(dev(o), dev(d), dev(area), dev(st), dev(en), times=dev(times), order=order)
    cnt = fld.hashgrid.prepare_actors(dev(o), dev(d), dev(area), dev(st), dev(en), dev(times))[1][0]
    assert int(cnt.max()) > 8 and int((cnt == 0).sum()) > 50
    assert rel_l2(host(feats), want_f) < TOL
    assert rel_l2(host(acc), want_a) < TOL
    assert rel_l2(host(depth), want_d) < TOL
    # early termination stays within its bound on the actor kernel too
    with torch.no_grad():
        f2, _, a2 = fld.render(dev(o), dev(d), dev(area), dev(st), dev(en), times=dev(times), early_stop_eps=1e-3)
    assert float((a2 - acc).abs().max()) <= 1e-3 + 1e-6 and float((f2 - feats).abs().max()) < 5e-2


def test_fused_render_with_actors_fp16_tables_equal_rounded_fp32_tables():
    """BASELINE config[4]'s storage: fp16 hash tables (static and actors).  The kernel converts entries on load, so the
    result must equal, bit for bit, the fp32 run on tables holding the rounded values."""
    g = load_golden("field_actors")
    outs = []
    for half in (False, True):
        fld = make_field()
        for grid in [fld.hashgrid.static_grid, *fld.hashgrid.actor_grids]:
            rounded = grid.hash_table.data.half()
            grid.hash_table.data = rounded if half else rounded.float()
        assert fld.fused_supported(with_actors=True)
        with torch.no_grad():
            outs.append(fld.render(dev(g["o"]), dev(g["d"]), dev(g["area"]), dev(g["starts"]), dev(g["ends"]),
                                   times=dev(g["times"])))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    # mixed storage is refused, not silently converted
    fld = make_field()
    fld.hashgrid.static_grid.hash_table.data = fld.hashgrid.static_grid.hash_table.data.half()
    assert not fld.fused_supported(with_actors=True)


@pytest.mark.parametrize("scale", [10.0, 1.5])
def test_actor_pair_positions_kernel_vs_torch_autograd(scale):
    """nrhip_actor_pair_positions_fwd/bwd (pose interpolation -> 6-D rotation -> inverse -> transform -> flip ->
    contraction, one thread per pair, hand-written backward) against the torch formulation of the same chain
    (model_components/dynamic_actors.world2box_pairs + the contraction ops; that formulation matches the reference's
    autograd, tests/golden/field_actors_grads).  scale 1.5 puts most pairs OUTSIDE the unit box of the contraction."""
    from neurad_studio_amd import autograd as ag
    from neurad_studio_amd.model_components.dynamic_actors import world2box_pairs

    g = load_golden("field_actors")
    fld = make_field()
    fld.hashgrid.config.actor.actor_scale = scale
    act = fld.hashgrid.actors
    o, d, a = dev(g["o"]), dev(g["d"]), dev(g["area"])
    st, en, times = dev(g["starts"]), dev(g["ends"]), dev(g["times"])
    R, S = st.shape
    torch.manual_seed(0)
    P = 4000
    idx = torch.randint(0, R * S, (P,), device="cuda")
    aidx = torch.randint(0, 3, (P,), device="cuda")
    flip = (torch.randint(0, 2, (R,), device="cuda") * 2 - 1).float()
    gx, gs = torch.randn(P, 3, device="cuda"), torch.randn(P, device="cuda")

    def torch_chain():
        ray, smp = idx // S, idx % S
        t0, t1 = st[ray, smp], en[ray, smp]
        dist = (t1 - t0) / 2
        t = t0 + dist
        mean = o[ray] + d[ray] * t[:, None]
        std = (a[ray] * t.pow(2) * dist).pow(1 / 3)
        r_inv, t_inv = world2box_pairs(act, times[ray], aidx)
        pos = (r_inv * mean[:, None, :]).sum(-1) + t_inv
        pos = torch.cat([pos[:, :1] * flip[ray, None], pos[:, 1:]], dim=-1)
        m, s = pos / scale, std / scale
        mag = m.abs().amax(dim=-1, keepdim=True)
        cm = mag.clamp_min(1.0)
        m = torch.where(mag < 1, m, (2 - 1 / cm) * (m / cm))
        s = torch.where(mag[:, 0] < 1, s, s * (((2 * cm[:, 0] - 1).pow(1 / 3) / cm[:, 0]) ** 2))
        return (m + 2.0) / 4.0, s / 4.0

    for p in (act.actor_positions, act.actor_rotations_6d):
        p.requires_grad_(True)
        p.grad = None
    x_ref, s_ref = torch_chain()
    ((x_ref * gx).sum() + (s_ref * gs).sum()).backward()
    want = [act.actor_positions.grad.clone(), act.actor_rotations_6d.grad.clone()]
    act.actor_positions.grad = act.actor_rotations_6d.grad = None
    x, s = ag.ActorPairPositionsFn.apply(act.actor_positions, act.actor_rotations_6d, fld.hashgrid.actor_spec(), o, d, a, st,
                                         en, times, idx, aidx.int(), flip)
    assert float((x - x_ref).abs().max()) < 2e-6 and rel_l2(host(s), host(s_ref)) < 1e-5
    if scale < 5:
        assert float(((x_ref - 0.5).abs().amax(-1) > 0.25).float().mean()) > 0.5  # the contracted branch is exercised
    ((x * gx).sum() + (s * gs).sum()).backward()
    assert rel_l2(host(act.actor_positions.grad), host(want[0])) < 1e-4
    assert rel_l2(host(act.actor_rotations_6d.grad), host(want[1])) < 1e-4


def test_actor_edits_vs_reference_golden():
    """Eval-time actor edit (DynamicActors.actor_editing -> nrhip_actor_prepare_edited) against the REFERENCE's outputs for
    the same edits (tests/golden/field_actors_edit.npz: a shift of every actor, a yaw of one, both on a clamped index, the
    ignored height-only edit): hit set, per-sample field outputs (operator path) and the fused render kernel; the edit is
    an eval-time thing -- the training mode ignores it, and asking for gradients of edited actors is refused."""
    from neurad_studio_amd import ops
    from neurad_studio_amd.cameras.rays import RayBundle
    from neurad_studio_amd.field_components.field_heads import FieldHeadNames

    g, ge = load_golden("field_actors"), load_golden("field_actors_edit")
    fld = make_field()
    R, S = g["starts"].shape
    rb = RayBundle(origins=dev(g["o"]), directions=dev(g["d"]), pixel_area=dev(g["area"])[:, None],
                   times=dev(g["times"])[:, None], nears=torch.zeros(R, 1, device="cuda"),
                   fars=torch.full((R, 1), 60.0, device="cuda"))
    rs = rb.get_ray_samples(dev(g["starts"])[..., None], dev(g["ends"])[..., None])
    o, d, a = rs.frustums.per_ray()
    act = fld.hashgrid.actors
    for e, (lat, lon, hgt, rot, idx) in enumerate(ge["edits"].tolist()):
        act.actor_editing.update(lateral=lat, longitudinal=lon, height=hgt, rotation=rot, index=idx)
        spec, cand = fld.hashgrid.prepare_actors(o, d, a, dev(g["starts"]), dev(g["ends"]), dev(g["times"]))
        _, hit = ops.actor_encode(spec, cand, o, d, a, dev(g["starts"]), dev(g["ends"]), torch.zeros((R * S, 32), device="cuda"))
        want = np.zeros((R, S), bool)
        want[ge[f"e{e}_hit_ray"], ge[f"e{e}_hit_sample"]] = True
        np.testing.assert_array_equal(host(hit).reshape(R, S) >= 0, want)
        with torch.no_grad():
            out = fld(rs)
            feats, depth, acc = fld.render(dev(g["o"]), dev(g["d"]), dev(g["area"]), dev(g["starts"]), dev(g["ends"]),
                                           times=dev(g["times"]))
        assert rel_l2(host(out[FieldHeadNames.ALPHA][..., 0]), ge[f"e{e}_alpha"]) < TOL
        has_f = f"e{e}_feature" in ge
        want_f, want_d, want_a = _composite_reference(ge[f"e{e}_feature"] if has_f else np.zeros((R, S, 32), np.float32),
                                                      ge[f"e{e}_alpha"], g["starts"], g["ends"])
        if has_f:
            assert rel_l2(host(out[FieldHeadNames.FEATURE]), ge[f"e{e}_feature"]) < TOL
            assert rel_l2(host(feats), want_f) < TOL
        assert rel_l2(host(acc), want_a) < TOL and rel_l2(host(depth), want_d) < TOL
    # a real edit moved something; in training mode the edit is not applied (dynamic_actors.py:261-265)
    act.actor_editing.update(lateral=1.5, longitudinal=-2.0, height=0.3, rotation=0.0, index=-1.0)
    assert not np.array_equal(ge["e0_alpha"], g["alpha"])
    fld.hashgrid.config.actor.flip_prob = 0.0
    fld.train()
    with torch.no_grad():
        out = fld(rs)
    assert rel_l2(host(out[FieldHeadNames.ALPHA][..., 0]), g["alpha"]) < TOL
    fld.eval()
    with pytest.raises(NotImplementedError, match="actor edits"):
        fld(rs)  # grad mode, eval, edited actors: the differentiable actor rows would ignore the edit


@pytest.mark.parametrize("n", [0, 1, 1023, 1024, 70001])
def test_actor_pairs_equal_nonzero_of_the_hits_table(n):
    """ops.actor_pairs: the (sample, containing actor) pairs in (sample, slot) order -- exactly what
    `(hits >= 0).nonzero()` followed by `hits[idx, slot]` gives, empty tables and ragged last blocks included"""
    from neurad_studio_amd import ops

    g = np.random.default_rng(n)
    hits = np.full((n, 8), -1, np.int32)
    for i in np.flatnonzero(g.uniform(size=n) < 0.07):  # ascending actors, compacted to the front (nrhip_actor_hits' layout)
        k = int(g.integers(1, 4))
        hits[i, :k] = np.sort(g.choice(32, k, replace=False))
    if n > 10:
        hits[n - 1, :8] = np.arange(8)  # a full row at the very end
    h = torch.from_numpy(hits).cuda()
    si, ai = ops.actor_pairs(h)
    pair = (h >= 0).nonzero()
    assert si.dtype == torch.int64 and ai.dtype == torch.int32
    assert torch.equal(si, pair[:, 0]) and torch.equal(ai, h[pair[:, 0], pair[:, 1]])
