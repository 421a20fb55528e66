"""Pin the CPU oracle (oracle/neurad_oracle.py) against outputs of the REFERENCE ITSELF
(tests/golden/*.npz, produced by oracle/make_golden.py from the reference's torch path)."""
import numpy as np
import pytest

import neurad_oracle as O
import synth
from conftest import load_golden, rel_l2

TOL = 1e-5  # oracle restates the same fp32 ops; the product bar (HIP vs oracle) is 1e-4 rel-L2

HASH_TAGS = ["c2small", "neurad", "prop", "tiny", "actor"]


@pytest.mark.parametrize("tag", HASH_TAGS)
def test_hashgrid_forward_and_indices(tag):
    g = load_golden(f"hashgrid_{tag}")
    L, mn, mx, lg, F = (int(v) for v in g["cfg"])
    sc = O.hash_scalings(L, mn, mx)
    np.testing.assert_array_equal(sc, g["scalings"])
    table = synth.hash_table(L * 2**lg, F, seed=11)
    idx, _ = O.hashgrid_corner_indices(g["x"], sc, 2**lg)
    np.testing.assert_array_equal(idx[..., 6], g["h_floor"])  # bit-exact integer work
    np.testing.assert_array_equal(idx[..., 0], g["h_ceil"])
    y = O.hashgrid_fwd(g["x"], table, sc, 2**lg)
    assert rel_l2(y, g["y"]) < 1e-6


@pytest.mark.parametrize("tag", HASH_TAGS)
def test_hashgrid_backward(tag):
    g = load_golden(f"hashgrid_{tag}")
    L, mn, mx, lg, F = (int(v) for v in g["cfg"])
    sc = O.hash_scalings(L, mn, mx)
    gt = O.hashgrid_bwd(g["x"], g["grad_out"], sc, 2**lg, L * 2**lg, F)
    ref = np.zeros_like(gt)
    ref[g["grad_table_nz_idx"]] = g["grad_table_nz"]
    assert rel_l2(gt, ref) < TOL


def test_scalings_full_size_configs():
    g = load_golden("scalings")
    for tag, (L, mn, mx) in {"c2": (16, 16, 1024), "neurad": (8, 32, 8192), "prop": (6, 128, 4096),
                             "actor": (4, 64, 1024), "neurader": (8, 64, 16384)}.items():
        np.testing.assert_array_equal(O.hash_scalings(L, mn, mx), g[tag])


def test_sh():
    g = load_golden("sh")
    assert rel_l2(O.sh_deg4(g["d01"]), g["y"]) < 1e-6


def _mlp_params(cfg):
    i, n, w, o = (int(v) for v in cfg)
    dims = [i] + [w] * (n - 1) + [o]
    ws, bs = [], []
    for k in range(n):
        wk, bk = synth.linear(dims[k + 1], dims[k], 100 + 10 * k)
        ws.append(wk), bs.append(bk)
    return ws, bs


@pytest.mark.parametrize("tag", ["geo64", "feat64", "geo32", "lidar"])
def test_mlp_fwd_bwd(tag):
    g = load_golden(f"mlp_{tag}")
    ws, bs = _mlp_params(g["cfg"])
    y, hidden = O.mlp_fwd(g["x"], ws, bs, return_hidden=True)
    assert rel_l2(y, g["y"]) < TOL
    dx, dws, dbs = O.mlp_bwd(hidden, ws, g["grad_out"])
    assert rel_l2(dx, g["dx"]) < TOL
    for k in range(len(ws)):
        assert rel_l2(dws[k], g[f"dw{k}"]) < TOL
        assert rel_l2(dbs[k], g[f"db{k}"]) < TOL


def test_contraction():
    g = load_golden("contraction")
    m, s = O.contract_gaussian(g["mean"], g["std"][:, 0], float(g["scale"]))
    assert rel_l2(m, g["cmean"]) < 1e-6
    assert rel_l2(s, g["cstd"][:, 0]) < 1e-6
    assert m.min() >= 0 and m.max() <= 1


def field_params(use_sdf=True):
    grid = O.GridParams(synth.hash_table(8 * 2**11, 4, seed=51, scale=0.5), 8, 32, 8192, 11)
    gw, gb, fw, fb = [], [], [], []
    for k, (o, i) in enumerate([(32, 32), (33, 32)]):
        w, b = synth.linear(o, i, 200 + 10 * k)
        gw.append(w), gb.append(b)
    for k, (o, i) in enumerate([(32, 48), (32, 32), (32, 32)]):
        w, b = synth.linear(o, i, 300 + 10 * k)
        fw.append(w), fb.append(b)
    return O.FieldParams(grid, 100.0, gw, gb, fw, fb, use_sdf=use_sdf)


@pytest.mark.parametrize("tag", ["sdf", "density"])
def test_field_forward(tag):
    g = load_golden(f"field_{tag}")
    mean, std = O.fast_isotropic_gaussian(g["o"], g["d"], g["area"], g["starts"], g["ends"])
    assert rel_l2(mean, g["gmean"]) < 1e-6 and rel_l2(std, g["gstd"]) < 1e-5
    p = field_params(use_sdf=(tag == "sdf"))
    out = O.field_fwd(p, g["o"], g["d"], g["area"], g["starts"], g["ends"])
    assert rel_l2(out["feature"], g["feature"]) < TOL
    if tag == "sdf":
        assert rel_l2(out["sdf"], g["sdf"]) < TOL
        assert rel_l2(out["alpha"], g["alpha"]) < TOL
    else:
        assert rel_l2(out["density"], g["density"]) < TOL


def test_field_forward_multisampled():
    """NeuRADFieldConfig.num_multisamples = 3 against the reference's own outputs (oracle/make_golden_multisample.py): the
    sub-sample gaussians, the averaged rescaled encoding, the field heads"""
    g, gm = load_golden("field_sdf"), load_golden("field_multisample")
    M = int(gm["num_multisamples"])
    st, en = g["starts"], g["ends"]
    step = ((en - st) / np.float32(M + 1)).astype(np.float32)
    for k in range(1, M + 1):  # sub-sample k IS the M = 1 gaussian of (t_k - step, t_k + step)
        tk = (st + np.float32(k) * step).astype(np.float32)
        mean, std = O.fast_isotropic_gaussian(g["o"], g["d"], g["area"], tk - step, tk + step)
        assert rel_l2(mean, gm["gmean"][:, :, k - 1]) < 1e-6 and rel_l2(std, gm["gstd"][:, :, k - 1]) < 1e-5
    p = field_params(use_sdf=True)
    enc = O.encode_static(p.grid, p.static_scale, g["o"], g["d"], g["area"], st, en, num_multisamples=M)
    assert rel_l2(enc, gm["enc"]) < TOL
    out = O.field_fwd(p, g["o"], g["d"], g["area"], st, en, num_multisamples=M)
    assert rel_l2(out["feature"], gm["feature"]) < TOL and rel_l2(out["sdf"], gm["sdf"]) < TOL
    assert rel_l2(out["alpha"], gm["alpha"]) < TOL and rel_l2(gm["alpha"], g["alpha"]) > 1e-3


def prop_params(seed, lg=11):
    w, _ = synth.linear(1, 6, seed + 1, bias=False)
    return O.ProposalParams(O.GridParams(synth.hash_table(6 * 2**lg, 1, seed=seed, scale=2.0), 6, 128, 4096, lg),
                            100.0, w + np.float32(0.3))


def test_ray_gradients_vs_reference_autograd():
    """dL/d(origins, directions) -- what a camera optimizer that moves the rays receives (cameras/camera_optimizers.py:173-182)
    -- of the static encoding and of a proposal density: the oracle's analytic chain against the reference's autograd
    (oracle/make_golden_raygrads.py)"""
    g = load_golden("ray_grads")
    p = field_params(use_sdf=True)
    go, gd = O.encode_static_ray_grads(p.grid, p.static_scale, g["o"], g["d"], g["area"], g["starts"], g["ends"], g["g_enc"])
    assert rel_l2(go, g["enc_go"]) < TOL and rel_l2(gd, g["enc_gd"]) < TOL
    pp = prop_params(91)
    assert rel_l2(O.proposal_density(pp, g["o"], g["d"], g["area"], g["starts"], g["ends"]), g["prop_dens"]) < TOL
    go, gd = O.proposal_density_ray_grads(pp, g["o"], g["d"], g["area"], g["starts"], g["ends"], g["prop_g_dens"])
    assert rel_l2(go, g["prop_go"]) < TOL and rel_l2(gd, g["prop_gd"]) < TOL
    # both regimes of the contraction are in the fixture, and the inf-norm's arg-max coordinate changes along some rays
    mean, _ = O.fast_isotropic_gaussian(g["o"], g["d"], g["area"], g["starts"], g["ends"])
    mag = np.abs(mean / np.float32(100.0)).max(-1)
    assert (mag < 1).any() and (mag > 1).any()
    assert (np.abs(mean).argmax(-1).max(1) != np.abs(mean).argmax(-1).min(1)).any()


def test_sampler_parts():
    g = load_golden("sampler_parts")
    R = g["o"].shape[0]
    bins, eu, sp = O.power_sampler(np.zeros(R), g["fars"], 128)
    assert rel_l2(bins, g["sp0"]) < 1e-6
    assert rel_l2(eu, g["eu0"]) < TOL
    p = prop_params(95)
    dens = O.proposal_density(p, g["o"], g["d"], g["area"], g["eu0"][:, :-1], g["eu0"][:, 1:])
    assert rel_l2(dens, g["dens0"]) < TOL
    w = O.weights_from_density(g["eu0"][:, 1:] - g["eu0"][:, :-1], g["dens0"])
    assert rel_l2(w, g["w0"]) < TOL
    nb, neu = O.pdf_sample(g["w0"], g["sp0"], 64, sp)
    assert rel_l2(nb, g["sp1"]) < TOL
    assert rel_l2(neu, g["eu1"]) < TOL


def test_sampler_chain_with_late_binding_quirk():
    g = load_golden("sampler_chain")
    R = g["o"].shape[0]
    props = [prop_params(91), prop_params(95)]
    out = O.proposal_sampler(props, g["o"], g["d"], g["area"], np.zeros(R), g["fars"], stretch_sky=False)
    for a, b in [(out.prop_weights[0], g["w0"]), (out.prop_weights[1], g["w1"]), (out.prop_starts[0], g["s0"]),
                 (out.prop_ends[1], g["e1"]), (out.starts, g["starts"]), (out.ends, g["ends"]),
                 (out.spacing_starts, g["sps"]), (out.spacing_ends, g["spe"])]:
        assert rel_l2(a, b) < 5e-5
    # without the quirk (round 0 on proposal_fields[0]) the result must differ -> the quirk is really pinned
    # M1 sky stretch (models/neurad.py:451-455): last end -> sky_distance, spacing_end -> 1-1e-7
    out3 = O.proposal_sampler(props, g["o"], g["d"], g["area"], np.zeros(R), g["fars"])
    assert np.all(out3.ends[:, -1] == np.float32(20000.0)) and np.all(out3.spacing_ends[:, -1] == np.float32(1 - 1e-7))
    np.testing.assert_array_equal(out3.ends[:, :-1], out.ends[:, :-1])
    out2 = O.proposal_sampler(props, g["o"], g["d"], g["area"], np.zeros(R), g["fars"], late_binding_quirk=False)
    assert rel_l2(out2.prop_weights[0], g["w0"]) > 1e-2


def test_sampler_train_mode_injected_jitter():
    g = load_golden("sampler_train")
    R = g["fars"].shape[0]
    bins, eu, sp = O.power_sampler(np.zeros(R), g["fars"], 128, t_rand=g["t_rand"])
    assert rel_l2(bins, g["sp0"]) < 1e-6 and rel_l2(eu, g["eu0"]) < TOL
    assert np.array_equal(bins, g["sp0"]) and np.array_equal(eu, g["eu0"])  # bit for bit: x ** -1 as ATen's reciprocal
    nb, neu = O.pdf_sample(g["w0"], g["sp0"], 64, sp, rand=g["rand1"])
    assert rel_l2(nb, g["sp1"]) < TOL and rel_l2(neu, g["eu1"]) < TOL


def test_compositing_restatement_consistency():
    """nerfacc is un-vendored and replaced by a placeholder on CPU (models/neurad.py:713-715): the
    compositing oracle is 'parity unpinned'.  What CAN be pinned: (a) the density variant equals the
    in-repo RaySamples.get_weights (golden w0), (b) the alpha variant agrees with the in-repo sibling
    get_weights_and_transmittance_from_alphas up to its +1e-7 epsilon."""
    g = load_golden("sampler_parts")
    s, e = g["eu0"][:, :-1], g["eu0"][:, 1:]
    w, trans, alphas = O.render_weight_from_density(s, e, g["dens0"])
    assert rel_l2(w, g["w0"]) < TOL
    ga = load_golden("weights_alpha_eps")
    wa, tr = O.render_weight_from_alpha(ga["alphas"])
    assert np.abs(wa - ga["w"]).max() < 5e-6
    # closed-form properties: sum(w) = 1 - prod(1-a); composite puts the residual on the sky sample
    acc = wa.sum(-1)
    np.testing.assert_allclose(acc, 1 - np.prod(1 - ga["alphas"].astype(np.float64), -1), rtol=0, atol=2e-6)
    feats = synth.normal(ga["alphas"].shape + (5,), seed=3)
    f, d, a = O.composite(wa, feats, np.zeros_like(wa), np.ones_like(wa))
    np.testing.assert_allclose(a[:, 0], acc, atol=1e-6)
    np.testing.assert_allclose(f, (wa[..., None] * feats).sum(1) + (1 - acc)[:, None] * feats[:, -1], atol=2e-6)


def test_sampler_losses_vs_reference_autograd():
    """zipnerf_interlevel_loss and distortion_loss (model_components/losses.py:137-156,645-705),
    values and gradients from the reference's own autograd (oracle/make_golden_losses.py)."""
    g = load_golden("losses")
    R = g["w0"].shape[0]
    total = 0.0
    for cp, wp, gw, r in [(g["sd0"], g["w0"], g["g_w0"], 0.03), (g["sd1"], g["w1"], g["g_w1"], 0.003)]:
        loss, _, grad = O.interlevel_loss_level(g["sdf"], g["wf"], cp, wp, r)
        total += loss.mean()
        assert rel_l2(grad / R, gw) < 1e-5
    assert abs(total - g["interlevel"]) / g["interlevel"] < 1e-5
    loss, grad = O.distortion_loss_rays(g["sdf"], g["wf"])
    assert abs(loss.mean() - g["distortion"]) / g["distortion"] < 1e-5 and rel_l2(grad / R, g["g_wf"]) < 1e-5


PATCH_CASES = ["neurad", "odd", "even", "single"]


@pytest.mark.parametrize("tag", PATCH_CASES)
def test_patch_sampler_restatement_vs_reference(tag):
    """oracle patch sampler == the reference's ScaledPatchSampler on the recorded torch.rand draws: integer work, bit-exact"""
    g = load_golden("patch_sampler")
    n, h, w, ps, sc = (int(v) for v in g[f"{tag}_shape"])
    c = O.patch_centers_from_uniforms(g[f"{tag}_uniforms"], n, h, w, ps * sc)
    rays, coords, patches = O.patches_from_centers(g[f"{tag}_image"], c, ps, sc, image_idx=g[f"{tag}_image_idx"])
    np.testing.assert_array_equal(rays, g[f"{tag}_indices"])
    np.testing.assert_array_equal(patches, g[f"{tag}_patches"])
    np.testing.assert_array_equal(coords, g[f"{tag}_coords"])


def test_patch_sampler_restatement_given_centers():
    g = load_golden("patch_sampler")
    n, h, w, ps, sc = (int(v) for v in g["centers_shape"])
    rays, _, patches = O.patches_from_centers(g["centers_image"], g["centers_centers"], ps, sc)
    np.testing.assert_array_equal(rays, g["centers_indices"])
    np.testing.assert_array_equal(patches, g["centers_patches"])


@pytest.mark.parametrize("tag,rays", [("lidar", 203), ("lidar_one", 16)])
def test_lidar_point_sampler_restatement_vs_reference(tag, rays):
    g = load_golden("patch_sampler")
    idx, pts = O.lidar_point_sample(g[f"{tag}_cloud"], g[f"{tag}_points_per_lidar"], rays, g[f"{tag}_shuffle"],
                                    g[f"{tag}_draws"], lidar_idx=g[f"{tag}_lidar_idx"])
    np.testing.assert_array_equal(idx, g[f"{tag}_indices"])
    np.testing.assert_array_equal(pts, g[f"{tag}_points"])
