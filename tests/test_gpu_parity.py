"""GPU parity tests: the hand-written HIP path (through the C ABI) vs the CPU oracle, plus the golden
vectors the reference itself produced.  Bar: bit-exact for hash indices (checked through exact lattice
lookups), <= 1e-4 rel-L2 for floating point (BASELINE.json north_star); most ops land near 1e-6."""
import numpy as np
import pytest
import torch

import neurad_oracle as O
import synth
from conftest import load_golden, rel_l2

pytestmark = pytest.mark.gpu

TOL = 1e-4  # north_star tolerance (rel-L2 vs the reference's fp32 torch path)
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


HASH_CFGS = {"c2small": (16, 16, 1024, 12, 2), "neurad": (8, 32, 8192, 12, 4), "prop": (6, 128, 4096, 11, 1),
             "tiny": (1, 32, 32, 10, 4), "actor": (4, 64, 1024, 10, 4)}


@pytest.mark.parametrize("tag", list(HASH_CFGS))
def test_hashgrid_fwd_vs_reference_golden(ops, tag):
    g = load_golden(f"hashgrid_{tag}")
    L, mn, mx, lg, F = HASH_CFGS[tag]
    spec = ops.GridSpec(L, F, lg, mn, mx)
    np.testing.assert_array_equal(spec.scalings.numpy(), g["scalings"])
    table = synth.hash_table(L * 2**lg, F, seed=11)
    y = host(ops.hashgrid_fwd(spec, dev(table), dev(g["x"])))
    assert rel_l2(y, g["y"]) < TIGHT  # fma contraction in the lerp tree vs torch's separate mul/add
    # fp16 table storage (BASELINE config 5): same kernel, half loads; compare with the oracle on the rounded table
    t16 = table.astype(np.float16)
    y16 = host(ops.hashgrid_fwd(spec, dev(t16, torch.float16), dev(g["x"])))
    ref16 = O.hashgrid_fwd(g["x"], t16.astype(np.float32), g["scalings"], 2**lg)
    assert rel_l2(y16, ref16) < TIGHT


def test_hash_indices_bit_exact(ops):
    """One-hot tables: entry r of level l set to 1 -> the output equals the sum of trilinear weights that hit
    r.  With x on lattice points (ceil == floor) the lookup returns exactly table[index] -> exact index check
    against the oracle's int64 hash over many random lattice points, incl. large coordinates."""
    L, mn, mx, lg, F = 8, 32, 8192, 14, 1
    spec = ops.GridSpec(L, F, lg, mn, mx)
    sc = spec.scalings.numpy()
    rng = np.random.default_rng(0)
    # table value = its own row index (exactly representable in fp32 below 2^24)
    table = (np.arange(L * 2**lg) % 2**lg).astype(np.float32)[:, None]
    lvl = 7
    pts = rng.integers(0, int(sc[lvl]) + 1, size=(4096, 3)).astype(np.float32) / sc[lvl]
    # keep points that are exact lattice points of level `lvl` after the fp32 multiply
    keep = np.all((pts * sc[lvl]) == np.round(pts * sc[lvl]), axis=1)
    pts = pts[keep]
    y = host(ops.hashgrid_fwd(spec, dev(table), dev(pts)))[:, lvl]
    idx, _ = O.hashgrid_corner_indices(pts, sc, 2**lg)
    np.testing.assert_array_equal(y.astype(np.int64), idx[:, lvl, 6] - lvl * 2**lg)


@pytest.mark.parametrize("atomic", [False, True], ids=["binned", "atomic"])
@pytest.mark.parametrize("tag", list(HASH_CFGS))
def test_hashgrid_bwd(ops, tag, atomic, monkeypatch):
    monkeypatch.setattr(ops, "_FORCE_ATOMIC_SCATTER", atomic)
    monkeypatch.setattr(ops, "_BINNED_MIN_SAMPLES", 1)  # the goldens are small batches
    g = load_golden(f"hashgrid_{tag}")
    L, mn, mx, lg, F = HASH_CFGS[tag]
    spec = ops.GridSpec(L, F, lg, mn, mx)
    gt = host(ops.hashgrid_bwd(spec, None, dev(g["x"]), dev(g["grad_out"])))
    ref = np.zeros_like(gt)
    ref[g["grad_table_nz_idx"]] = g["grad_table_nz"]
    assert rel_l2(gt, ref) < TIGHT


def test_sh4(ops):
    g = load_golden("sh")
    assert rel_l2(host(ops.sh4_fwd(dev(g["d01"]))), g["y"]) < 1e-6


def _mlp_params(cfg):
    i, n, w, o = (int(v) for v in cfg)
    dims = [i] + [w] * (n - 1) + [o]
    ws, bs = [], []
    for k in range(n):
        wk, bk = synth.linear(dims[k + 1], dims[k], 100 + 10 * k)
        ws.append(wk), bs.append(bk)
    return ws, bs


@pytest.mark.parametrize("tag", ["geo64", "feat64", "geo32", "lidar"])
def test_mlp_fwd_bwd_vs_reference_golden(ops, tag):
    g = load_golden(f"mlp_{tag}")
    ws, bs = _mlp_params(g["cfg"])
    dws, dbs = [dev(w) for w in ws], [dev(b) for b in bs]
    x = dev(g["x"])
    y, hidden = ops.mlp_fwd(x, dws, dbs, save_hidden=True)
    assert rel_l2(host(y), g["y"]) < TIGHT
    gx, gws, gbs = ops.mlp_bwd(x, hidden, dev(g["grad_out"]), dws, dbs)
    assert rel_l2(host(gx), g["dx"]) < TIGHT
    for k in range(len(ws)):
        assert rel_l2(host(gws[k]), g[f"dw{k}"]) < TIGHT, k
        assert rel_l2(host(gbs[k]), g[f"db{k}"]) < TIGHT, k


@pytest.mark.parametrize("dims", [(3, 1, 7, 5), (13, 4, 24, 3), (48, 3, 128, 16), (32, 2, 16, 1), (5, 2, 100, 9)])
def test_mlp_odd_shapes_and_ragged_batches(ops, dims):
    i, n, w, o = dims
    dd = [i] + [w] * (n - 1) + [o]
    ws, bs = [], []
    for k in range(n):
        wk, bk = synth.linear(dd[k + 1], dd[k], 500 + k)
        ws.append(wk), bs.append(bk if k % 2 == 0 else None)
    for N in (1, 15, 16, 17, 1000):
        x = synth.normal((N, i), seed=N)
        y = host(ops.mlp_fwd(dev(x), [dev(a) for a in ws], [None if b is None else dev(b) for b in bs]))
        ref = O.mlp_fwd(x, ws, bs)
        assert rel_l2(y, ref) < TIGHT, (dims, N)
    assert ops.mlp_fwd(dev(np.zeros((0, i), np.float32)), [dev(a) for a in ws],
                       [None if b is None else dev(b) for b in bs]).shape == (0, o)


def field_params(use_sdf=True, L=8, F=4, lg=11, H=32, mn=32, mx=8192, scale=0.5):
    grid = O.GridParams(synth.hash_table(L * 2**lg, F, seed=51, scale=scale), L, mn, mx, lg)
    gw, gb, fw, fb = [], [], [], []
    for k, (o, i) in enumerate([(H, 32), (33, H)]):
        w, b = synth.linear(o, i, 200 + 10 * k)
        gw.append(w), gb.append(b)
    for k, (o, i) in enumerate([(H, 48), (H, H), (32, H)]):
        w, b = synth.linear(o, i, 300 + 10 * k)
        fw.append(w), fb.append(b)
    return O.FieldParams(grid, 100.0, gw, gb, fw, fb, use_sdf=use_sdf)


def to_spec(ops, p: O.FieldParams, half=False):
    g = p.grid
    spec = ops.GridSpec(g.num_levels, g.n_feat, g.log2_hashmap_size, g.min_res, g.max_res)
    table = dev(g.table, torch.float16 if half else torch.float32)
    return ops.FieldSpec(spec, table, p.static_scale, [dev(w) for w in p.geo_w], [dev(b) for b in p.geo_b],
                         [dev(w) for w in p.feat_w], [dev(b) for b in p.feat_b], use_sdf=p.use_sdf,
                         beta=abs(p.beta) + p.beta_min)


@pytest.mark.parametrize("tag", ["sdf", "density"])
def test_field_fwd_vs_reference_golden(ops, tag):
    g = load_golden(f"field_{tag}")
    p = field_params(use_sdf=(tag == "sdf"))
    fs = to_spec(ops, p)
    feat, sdf, head = ops.field_fwd(fs, dev(g["o"]), dev(g["d"]), dev(g["area"]), dev(g["starts"]), dev(g["ends"]))
    assert rel_l2(host(feat), g["feature"]) < TOL
    if tag == "sdf":
        assert rel_l2(host(sdf), g["sdf"]) < TOL
        assert rel_l2(host(head), g["alpha"]) < TOL
    else:
        assert rel_l2(host(head), g["density"]) < TOL
    # the unfused encode op (H2-H4) against the oracle
    enc = host(ops.encode_fwd(fs.grid, fs.table, 100.0, dev(g["o"]), dev(g["d"]), dev(g["area"]), dev(g["starts"]),
                              dev(g["ends"])))
    ref = O.encode_static(p.grid, 100.0, g["o"], g["d"], g["area"], g["starts"], g["ends"])
    assert rel_l2(enc, ref) < TIGHT


def _sample_rays(R, S, seed, fars=200.0):
    o, d, area, _ = synth.rays(R, seed)
    bins, eu, _ = O.power_sampler(np.zeros(R), np.full(R, fars, np.float32), S)
    return o, d, area, np.ascontiguousarray(eu[:, :-1]), np.ascontiguousarray(eu[:, 1:]), eu


RENDER_CFGS = [  # (L, F, lg, min_res, max_res, H, use_sdf, R, S)
    (16, 2, 12, 16, 1024, 64, True, 37, 128),   # BASELINE config 2 shape (small table)
    (8, 4, 11, 32, 8192, 32, True, 50, 32),     # NeuRAD defaults
    (8, 4, 11, 32, 8192, 32, False, 21, 33),    # density head, ragged S (not a multiple of 16)
    (16, 2, 12, 16, 1024, 32, True, 5, 7),      # S < 16
    (8, 4, 11, 32, 8192, 64, False, 9, 1),      # single sample per ray
    (4, 8, 10, 64, 1024, 64, True, 13, 48),
]


@pytest.mark.parametrize("cfg", RENDER_CFGS)
def test_render_fused_vs_oracle(ops, cfg):
    L, F, lg, mn, mx, H, use_sdf, R, S = cfg
    p = field_params(use_sdf=use_sdf, L=L, F=F, lg=lg, H=H, mn=mn, mx=mx, scale=2.0 if use_sdf else 0.5)
    if use_sdf:
        p.beta = 3.0  # keep alphas away from saturation so the compositing is exercised
    fs = to_spec(ops, p)
    o, d, area, s, e, eu = _sample_rays(R, S, seed=R + S)
    ref = O.render_rays(p, o, d, area, s, e)
    edges = dev(eu)
    # bin EDGES passed as strided views: starts = edges[:, :-1], ends = edges[:, 1:]  (no copies)
    feats, depth, acc, w = ops.render_fwd(fs, dev(o), dev(d), dev(area), edges[:, :-1], edges[:, 1:],
                                          return_weights=True)
    assert rel_l2(host(w), ref["weights"]) < TOL
    assert rel_l2(host(feats), ref["features"]) < TOL
    assert rel_l2(host(depth), ref["depth"]) < TOL or np.abs(host(depth) - ref["depth"]).max() < 1e-5
    assert rel_l2(host(acc), ref["accumulation"]) < TOL
    # per-sample variant (Field.forward boundary) and the unfused C1+C2 ops composed by hand
    f2, sdf2, head2 = ops.field_fwd(fs, dev(o), dev(d), dev(area), dev(s), dev(e))
    assert rel_l2(host(f2), ref["feature"]) < TOL
    if use_sdf:
        w2, _ = ops.render_weight_from_alpha(head2)
    else:
        w2, _, _ = ops.render_weight_from_density(dev(s), dev(e), head2)
    cf, cd, ca = ops.composite_fwd(w2, f2, dev(s), dev(e))
    assert rel_l2(host(cf), host(feats)) < 1e-5 and rel_l2(host(ca), host(acc)) < 1e-5


def test_render_fp16_table(ops):
    p = field_params(use_sdf=True, L=8, F=4, lg=11, H=32)
    p.beta = 3.0
    p.grid.table = p.grid.table.astype(np.float16).astype(np.float32)  # oracle sees the rounded values
    fs = to_spec(ops, p, half=True)
    o, d, area, s, e, _ = _sample_rays(33, 32, seed=5)
    ref = O.render_rays(p, o, d, area, s, e)
    feats, depth, acc = ops.render_fwd(fs, dev(o), dev(d), dev(area), dev(s), dev(e))
    assert rel_l2(host(feats), ref["features"]) < TOL and rel_l2(host(acc), ref["accumulation"]) < TOL


def test_compositing_ops_fwd_bwd(ops):
    R, S, Cc = 19, 70, 32  # S > 64 exercises the carried scan
    alphas = synth.uniform((R, S), 0.0, 0.2, seed=1)
    feats = synth.normal((R, S, Cc), seed=2)
    s = np.sort(synth.uniform((R, S + 1), 0.0, 50.0, seed=3), -1)
    st, en = np.ascontiguousarray(s[:, :-1]), np.ascontiguousarray(s[:, 1:])
    w, t = ops.render_weight_from_alpha(dev(alphas))
    rw, rt = O.render_weight_from_alpha(alphas)
    assert rel_l2(host(w), rw) < TIGHT and rel_l2(host(t), rt) < TIGHT
    sig = synth.uniform((R, S), 0.0, 0.3, seed=4)
    w2, t2, a2 = ops.render_weight_from_density(dev(st), dev(en), dev(sig))
    rw2, rt2, ra2 = O.render_weight_from_density(st, en, sig)
    assert rel_l2(host(w2), rw2) < TIGHT and rel_l2(host(t2), rt2) < TIGHT and rel_l2(host(a2), ra2) < TIGHT
    assert rel_l2(host(ops.weights_from_density(dev(en - st), dev(sig))), O.weights_from_density(en - st, sig)) < TIGHT
    for Cv in (1, 3, 32, 48):
        v = synth.normal((R, S, Cv), seed=5 + Cv)
        assert rel_l2(host(ops.accumulate_along_rays(dev(rw), dev(v))), O.accumulate_along_rays(rw, v)) < TIGHT
    assert rel_l2(host(ops.accumulate_along_rays(dev(rw))), O.accumulate_along_rays(rw)) < TIGHT
    of, od, oa = ops.composite_fwd(dev(rw), dev(feats), dev(st), dev(en))
    rf, rd, ra = O.composite(rw, feats, st, en)
    assert rel_l2(host(of), rf) < TIGHT and rel_l2(host(od), rd) < TIGHT and rel_l2(host(oa), ra) < TIGHT
    # backward vs torch autograd of the same dense formulas (fp64 on CPU)
    ta = torch.tensor(alphas, dtype=torch.float64, requires_grad=True)
    tf = torch.tensor(feats, dtype=torch.float64, requires_grad=True)
    trans = torch.cumprod(torch.cat([torch.ones(R, 1, dtype=torch.float64), 1 - ta[:, :-1]], -1), -1)
    tw = trans * ta
    acc = tw.sum(-1, keepdim=True)
    tw2 = torch.cat([tw[:, :-1], tw[:, -1:] + 1 - acc], -1)
    tfeat = (tw2[..., None] * tf).sum(1)
    mid = torch.tensor((st + en) / 2, dtype=torch.float64)
    tdepth = (tw2[:, :-1] * mid[:, :-1]).sum(-1, keepdim=True)
    gF, gD, gA = synth.normal((R, Cc), 7), synth.normal((R, 1), 8), synth.normal((R, 1), 9)
    (tfeat * torch.tensor(gF)).sum().add((tdepth * torch.tensor(gD)).sum()).add((acc * torch.tensor(gA)).sum()).backward()
    gw, gf = ops.composite_bwd(dev(rw), dev(feats), dev(st), dev(en), dev(gF), dev(gD), dev(gA))
    ga = ops.render_weight_from_alpha_bwd(dev(alphas), gw)
    assert rel_l2(host(gf), tf.grad.numpy()) < TIGHT
    assert rel_l2(host(ga), ta.grad.numpy()) < TOL
    # density-mode backward
    tsig = torch.tensor(sig, dtype=torch.float64, requires_grad=True)
    dl = torch.tensor(en - st, dtype=torch.float64)
    sd = tsig * dl
    tr = torch.exp(-torch.cat([torch.zeros(R, 1, dtype=torch.float64), torch.cumsum(sd[:, :-1], -1)], -1))
    twd = (1 - torch.exp(-sd)) * tr
    gwd = synth.normal((R, S), 11)
    (twd * torch.tensor(gwd)).sum().backward()
    gs = ops.render_weight_from_density_bwd(dev(st), dev(en), dev(sig), dev(gwd))
    assert rel_l2(host(gs), tsig.grad.numpy()) < TOL
    gs2 = ops.weights_from_density_bwd(dev(en - st), dev(sig), dev(gwd))
    assert rel_l2(host(gs2), tsig.grad.numpy()) < TOL


def prop_params(seed, lg=11):
    w, _ = synth.linear(1, 6, seed + 1, bias=False)
    return O.ProposalParams(O.GridParams(synth.hash_table(6 * 2**lg, 1, seed=seed, scale=2.0), 6, 128, 4096, lg),
                            100.0, w + np.float32(0.3))


def to_pspec(ops, p):
    g = p.grid
    return ops.ProposalSpec(ops.GridSpec(g.num_levels, 1, g.log2_hashmap_size, g.min_res, g.max_res), dev(g.table),
                            p.static_scale, dev(p.decoder_w))


def test_sampler_pieces_vs_reference_golden(ops):
    g = load_golden("sampler_parts")
    R = g["o"].shape[0]
    sp0, eu0 = ops.power_sampler(None, dev(g["fars"]), 128)
    assert rel_l2(host(sp0), g["sp0"]) < 1e-6 and rel_l2(host(eu0), g["eu0"]) < TOL
    # PowerSampler's bins are the reference's bit for bit since the exponent -1 is evaluated the way ATen does (a reciprocal)
    assert np.array_equal(host(sp0), g["sp0"]) and np.array_equal(host(eu0), g["eu0"])
    ps = to_pspec(ops, prop_params(95))
    e0 = dev(g["eu0"])
    dens = ops.proposal_density_fwd(ps, dev(g["o"]), dev(g["d"]), dev(g["area"]), e0[:, :-1], e0[:, 1:])
    assert rel_l2(host(dens), g["dens0"]) < TOL
    w = ops.weights_from_density(dev(g["eu0"][:, 1:] - g["eu0"][:, :-1]), dev(g["dens0"]))
    assert rel_l2(host(w), g["w0"]) < TIGHT
    sp1, eu1 = ops.pdf_sample(dev(g["w0"]), dev(g["sp0"]), None, dev(g["fars"]), 64)
    assert rel_l2(host(sp1), g["sp1"]) < TOL and rel_l2(host(eu1), g["eu1"]) < TOL
    # training-mode jitter, injected
    gt = load_golden("sampler_train")
    sp0t, eu0t = ops.power_sampler(None, dev(gt["fars"]), 128, t_rand=dev(gt["t_rand"]))
    assert rel_l2(host(sp0t), gt["sp0"]) < 1e-6 and rel_l2(host(eu0t), gt["eu0"]) < TOL
    assert np.array_equal(host(sp0t), gt["sp0"]) and np.array_equal(host(eu0t), gt["eu0"])
    sp1t, eu1t = ops.pdf_sample(dev(gt["w0"]), dev(gt["sp0"]), None, dev(gt["fars"]), 64, rand=dev(gt["rand1"]))
    assert rel_l2(host(sp1t), gt["sp1"]) < TOL and rel_l2(host(eu1t), gt["eu1"]) < TOL


def test_fused_proposal_sampler_vs_reference_golden(ops):
    g = load_golden("sampler_chain")
    props = [prop_params(91), prop_params(95)]
    # the reference's late-binding closure evaluates proposal_fields[1] in BOTH rounds (models/neurad.py:248)
    specs = [to_pspec(ops, props[1]), to_pspec(ops, props[1])]
    ws, sps, eus = ops.proposal_sampler_fwd(specs, dev(g["o"]), dev(g["d"]), dev(g["area"]), None, dev(g["fars"]))
    assert rel_l2(host(ws[0]), g["w0"]) < TOL and rel_l2(host(ws[1]), g["w1"]) < TOL
    assert rel_l2(host(eus[0][:, :-1]), g["s0"]) < TOL and rel_l2(host(eus[1][:, 1:]), g["e1"]) < TOL
    assert rel_l2(host(eus[2][:, :-1]), g["starts"]) < TOL and rel_l2(host(eus[2][:, 1:]), g["ends"]) < TOL
    assert rel_l2(host(sps[2][:, :-1]), g["sps"]) < TOL and rel_l2(host(sps[2][:, 1:]), g["spe"]) < TOL


@pytest.mark.parametrize("atomic", [False, True], ids=["binned", "atomic"])
def test_proposal_density_bwd(ops, atomic, monkeypatch):
    monkeypatch.setattr(ops, "_FORCE_ATOMIC_SCATTER", atomic)
    monkeypatch.setattr(ops, "_BINNED_MIN_SAMPLES", 1)
    p = prop_params(95)
    ps = to_pspec(ops, p)
    R, S = 11, 40
    o, d, area, s, e, _ = _sample_rays(R, S, seed=3)
    dens = ops.proposal_density_fwd(ps, dev(o), dev(d), dev(area), dev(s), dev(e))
    gd = synth.normal((R, S), 12)
    gt, gdec = ops.proposal_density_bwd(ps, dev(o), dev(d), dev(area), dev(s), dev(e), dens, dev(gd))
    # oracle: d/dtable and d/ddecoder of sum(gd * exp(enc @ w))
    enc = O.encode_static(p.grid, p.static_scale, o, d, area, s, e).astype(np.float64)
    x = enc @ p.decoder_w.astype(np.float64).T
    gx = gd.reshape(-1, 1) * np.exp(np.clip(x, -15, 15))
    ref_dec = (gx * enc).sum(0)
    assert rel_l2(host(gdec)[0], ref_dec) < TOL
    mean, std = O.fast_isotropic_gaussian(o, d, area, s, e)
    pos, cstd = O.contract_gaussian(mean, std, p.static_scale)
    sc = p.grid.scalings
    rw = 1.0 / np.maximum(sc[None, :] * 2 * cstd.reshape(-1, 1), 1.0)
    g_enc = (gx * p.decoder_w.astype(np.float64)) * rw
    ref_t = O.hashgrid_bwd(pos.reshape(-1, 3), g_enc, sc, p.grid.table_size, 6 * p.grid.table_size, 1)
    assert rel_l2(host(gt), ref_t) < TOL
    # training path: the forward saves the rescaled per-level features, the decoder gradient streams them back
    dens2, lf = ops.proposal_density_fwd(ps, dev(o), dev(d), dev(area), dev(s), dev(e), save_features=True)
    assert rel_l2(host(dens2), host(dens)) < 1e-6 and rel_l2(host(lf).T, enc) < TIGHT  # lf is level-major [L, N]
    gt2, gdec2 = ops.proposal_density_bwd(ps, dev(o), dev(d), dev(area), dev(s), dev(e), dens, dev(gd),
                                          level_features=lf)
    assert rel_l2(host(gdec2)[0], ref_dec) < TOL and rel_l2(host(gt2), ref_t) < TOL


def test_empty_batches(ops):
    p = field_params()
    fs = to_spec(ops, p)
    z = lambda *s: torch.zeros(*s, device="cuda")  # noqa: E731
    f, dpt, a = ops.render_fwd(fs, z(0, 3), z(0, 3), z(0), z(0, 32), z(0, 32))
    assert f.shape == (0, 32) and a.shape == (0, 1)
    assert ops.hashgrid_fwd(fs.grid, fs.table, z(0, 3)).shape == (0, 32)


def test_full_size_properties_config2(ops):
    """BASELINE config 2 at full size (4096 x 128, L=16, T=2^19, 64-wide): size-independent properties.
    (a) accumulation == 1 - prod(1 - alpha) and sum of returned weights, (b) features are an affine function of
    the table restricted to what the MLP sees -> permuting rays permutes outputs, (c) fused == unfused ops."""
    L, F, lg, H, R, S = 16, 2, 19, 64, 4096, 128
    p = field_params(use_sdf=True, L=L, F=F, lg=lg, H=H, mn=16, mx=1024, scale=1.0)
    p.beta = 2.0
    fs = to_spec(ops, p)
    o, d, area, s, e, eu = _sample_rays(R, S, seed=77)
    do, dd, da, edges = dev(o), dev(d), dev(area), dev(eu)
    feats, depth, acc, w = ops.render_fwd(fs, do, dd, da, edges[:, :-1], edges[:, 1:], return_weights=True)
    f2, sdf2, alpha2 = ops.field_fwd(fs, do, dd, da, edges[:, :-1], edges[:, 1:])
    assert torch.isfinite(feats).all() and torch.isfinite(f2).all()
    pa = 1 - torch.prod(1 - alpha2.double(), -1)
    assert (acc[:, 0].double() - pa).abs().max() < 1e-5
    assert (w.sum(-1).double() - pa).abs().max() < 1e-5
    w2, _ = ops.render_weight_from_alpha(alpha2)
    cf, cd, ca = ops.composite_fwd(w2, f2, edges[:, :-1].contiguous(), edges[:, 1:].contiguous())
    assert rel_l2(host(cf), host(feats)) < 1e-5 and rel_l2(host(cd), host(depth)) < 1e-5
    perm = torch.randperm(R, device="cuda")
    fp, dp, ap = ops.render_fwd(fs, do[perm], dd[perm], da[perm], edges[perm][:, :-1], edges[perm][:, 1:])
    assert torch.equal(fp, feats[perm]) and torch.equal(ap, acc[perm])
    # oracle on a slice the CPU finishes quickly
    sl = slice(100, 116)
    ref = O.render_rays(p, o[sl], d[sl], area[sl], s[sl], e[sl])
    assert rel_l2(host(feats[sl]), ref["features"]) < TOL and rel_l2(host(acc[sl]), ref["accumulation"]) < TOL


@pytest.mark.parametrize("cfg", [(16, 2, 19, 4096, 128), (8, 4, 16, 257, 33), (4, 8, 12, 64, 7), (6, 1, 20, 300, 48),
                                 (2, 4, 9, 5, 3), (6, 1, 14, 9001, 128),  # > 64 chunks: the two-level chunk prefix
                                 (6, 1, 14, 9001, 128, 18), (8, 4, 16, 2051, 33, 15)])  # rounds of 2^18 / 2^15 samples
def test_encode_bwd_binned_equals_atomic_scatter(ops, cfg, monkeypatch, switches):
    """B1 table gradient: the owner-computes path (LDS slices, no memory-side atomics) against the atomic
    scatter-add, from BASELINE config 2 at full size down to ragged batches and tables smaller than one slice.
    Same terms, different summation order -> agreement to fp32 rounding; linearity in grad_out is exact-ish too.
    A sixth entry sets NRHIP_BIN_ROUND_LOG2: the batch then goes through in several rounds."""
    if len(cfg) == 6:
        switches.set("NRHIP_BIN_ROUND_LOG2", str(cfg[5]))
    L, F, lg, R, S = cfg[:5]
    spec = ops.GridSpec(L, F, lg, 16, 2048)
    o, d, area, s, e, eu = _sample_rays(R, S, seed=5)
    do, dd, da, edges = dev(o), dev(d), dev(area), dev(eu)
    go = dev(synth.normal((R * S, L * F), 11))
    st, en = edges[:, :-1], edges[:, 1:]
    monkeypatch.setattr(ops, "_FORCE_ATOMIC_SCATTER", False)
    monkeypatch.setattr(ops, "_BINNED_MIN_SAMPLES", 1)
    binned = ops.encode_bwd(spec, 1.0, do, dd, da, st, en, go)
    binned2 = ops.encode_bwd(spec, 1.0, do, dd, da, st, en, 2 * go)
    again = ops.encode_bwd(spec, 1.0, do, dd, da, st, en, go)
    bad = go.clone()
    bad[0, 0], bad[-1, -1] = float("inf"), float("nan")
    poisoned = ops.encode_bwd(spec, 1.0, do, dd, da, st, en, bad)
    monkeypatch.setattr(ops, "_FORCE_ATOMIC_SCATTER", True)
    atomic = ops.encode_bwd(spec, 1.0, do, dd, da, st, en, go)
    assert torch.isfinite(binned).all()
    assert rel_l2(host(binned), host(atomic)) < 2e-6
    assert (binned - atomic).abs().max() <= 1e-5 * atomic.abs().max()
    assert rel_l2(host(binned2), 2 * host(binned)) < 2e-6
    # bit-reproducible (integer accumulation is associative), and non-finite gradients still poison their entries
    assert torch.equal(binned, again)
    assert not torch.isfinite(poisoned).all() and torch.isfinite(poisoned).float().mean() > 0.5
    # mass conservation: trilinear weights sum to 1, so each feature column's gradient mass is preserved
    rw_go = host(binned).reshape(L, -1, F).sum(1)
    assert np.allclose(rw_go, host(atomic).reshape(L, -1, F).sum(1), rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("mode", ["0", "1"], ids=["ray-major", "sample-index-major"])
@pytest.mark.parametrize("cfg", [(6, 1, 14, 128), (6, 1, 14, 64), (8, 4, 15, 32), (8, 4, 15, 16), (16, 2, 14, 64)])
def test_table_gradient_on_coherent_chunks_every_walk(ops, cfg, mode, monkeypatch, switches):
    """Camera-patch rays (one origin, directions a fraction of a degree apart): `prep` then walks a 4096-sample chunk
    sample-index-major (a 16-lane row = 16 neighbouring rays at one sample index) instead of ray-major
    (NRHIP_BIN_TRANSPOSE = 1 / 0).  Either walk sends the same terms: the result equals the atomic scatter-add, with silent
    samples (opaque tails, scattered zeros, whole silent rays) and a ragged last chunk, for the encode path and the
    proposal-density path.  (The coherent walk had no test of its own before round 5; the round's third walk, quads, was
    held to this test too before it was measured slower and reverted: profiles/r05_quad_walk_rejected.diff.)"""
    L, F, lg, S = cfg
    switches.set("NRHIP_BIN_TRANSPOSE", mode)
    spec = ops.GridSpec(L, F, lg, 16, 2048)
    R = 4096 // S * 5 + 7  # five coherent chunks and a ragged one
    o = np.tile(np.array([[1.5, -2.0, 0.7]], np.float32), (R, 1))
    ang = (np.arange(R, dtype=np.float32) * 2e-4)[:, None]
    d = np.concatenate([np.cos(ang), np.sin(ang), 0.05 + 0.3 * ang], -1)
    d = (d / np.linalg.norm(d, axis=-1, keepdims=True)).astype(np.float32)
    area = np.full((R,), 2.4e-6, np.float32)
    _, eu, _ = O.power_sampler(np.zeros(R), np.full(R, 60.0, np.float32), S)
    g = synth.normal((R, S, L * F), 23)
    cut = (S * (0.35 + 0.6 * synth.uniform((R,), 0, 1, 24))).astype(np.int64)  # opaque from sample cut[r] on: whole silent quads
    g[np.arange(S)[None, :] >= cut[:, None]] = 0.0
    g[synth.uniform((R, S), 0, 1, 25) < 0.15] = 0.0  # scattered silent samples inside live quads
    g[5::11] = 0.0  # whole rays
    do, dd, da, edges = dev(o), dev(d), dev(area), dev(eu)
    st, en = edges[:, :-1], edges[:, 1:]
    go = dev(g.reshape(R * S, L * F))
    monkeypatch.setattr(ops, "_FORCE_ATOMIC_SCATTER", False)
    monkeypatch.setattr(ops, "_BINNED_MIN_SAMPLES", 1)
    binned = ops.encode_bwd(spec, 1.0, do, dd, da, st, en, go)
    assert torch.equal(binned, ops.encode_bwd(spec, 1.0, do, dd, da, st, en, go))  # bit-reproducible
    monkeypatch.setattr(ops, "_FORCE_ATOMIC_SCATTER", True)
    atomic = ops.encode_bwd(spec, 1.0, do, dd, da, st, en, go)
    assert float(atomic.abs().max()) > 0
    assert rel_l2(host(binned), host(atomic)) < 2e-6
    assert (binned - atomic).abs().max() <= 1e-5 * atomic.abs().max()
    if F == 1:
        ps = ops.ProposalSpec(spec, dev(synth.hash_table(L << lg, 1, seed=3, scale=0.5)), 1.0, dev(synth.normal((1, L), 5)))
        dens = ops.proposal_density_fwd(ps, do, dd, da, st, en)
        gd = dev(g[..., 0].copy())
        monkeypatch.setattr(ops, "_FORCE_ATOMIC_SCATTER", False)
        gt, gdec = ops.proposal_density_bwd(ps, do, dd, da, st, en, dens, gd)
        monkeypatch.setattr(ops, "_FORCE_ATOMIC_SCATTER", True)
        gt2, gdec2 = ops.proposal_density_bwd(ps, do, dd, da, st, en, dens, gd)
        assert rel_l2(host(gt), host(gt2)) < 2e-6 and rel_l2(host(gdec), host(gdec2)) < 1e-5


@pytest.mark.parametrize("F", [1, 4])
def test_table_gradient_x_pairs_that_straddle_two_slices(ops, F, monkeypatch):
    """The radix partition sends one record per (floor x, ceil x) corner pair.  The two entries differ by
    floor x ^ ceil x, which reaches the slice bits only where floor x = k * 2^log2TS - 1: such a pair goes out as two
    records.  Here most samples sit in exactly those cell columns of the finest levels (and the rest anywhere); the result
    must equal the atomic scatter-add, bit-reproducibly."""
    L, lg = 4, 14
    spec = ops.GridSpec(L, F, lg, 256, 4096)  # slices of 2^9..2^12 entries < the finer levels' resolutions
    n = 40000
    g = np.random.default_rng(3)
    x = g.uniform(0.02, 0.98, (n, 3)).astype(np.float32)
    scal = spec.scalings.numpy()
    for l in range(L):  # a quarter of the samples per level: floor(x * scale_l) = 2^k - 1 for a random k >= 9
        rows = np.arange(l, n, 2 * L)
        k = g.integers(9, 12, rows.size)
        cell = (1 << k) - 1
        cell = np.minimum(cell, int(scal[l]) - 2)
        x[rows, 0] = ((cell + g.uniform(0.1, 0.9, rows.size)) / scal[l]).astype(np.float32)
    xs, go = dev(x), dev(synth.normal((n, L * F), 17))
    monkeypatch.setattr(ops, "_FORCE_ATOMIC_SCATTER", False)
    monkeypatch.setattr(ops, "_BINNED_MIN_SAMPLES", 1)
    binned = ops.hashgrid_bwd(spec, None, xs, go)
    again = ops.hashgrid_bwd(spec, None, xs, go)
    monkeypatch.setattr(ops, "_FORCE_ATOMIC_SCATTER", True)
    atomic = ops.hashgrid_bwd(spec, None, xs, go)
    assert torch.equal(binned, again)
    assert rel_l2(host(binned), host(atomic)) < 2e-6
    assert (binned - atomic).abs().max() <= 1e-5 * atomic.abs().max()


@pytest.mark.parametrize("cfg", [(16, 2, 19, 2048, 64), (8, 4, 16, 257, 33), (6, 1, 14, 300, 48)])
def test_table_gradient_skips_exactly_zero_samples_exactly(ops, cfg, monkeypatch):
    """Samples whose incoming gradient is exactly zero (the tail of a ray behind an opaque surface; scattered ones; whole
    rays) send no records.  The result must equal the atomic scatter-add of the same gradient, and -- integer
    accumulation -- must not change by a single bit when the silent samples' rows hold -0.0 instead of +0.0 or when
    silent samples sit between two samples of the same cell (they split a merged run, nothing else)."""
    L, F, lg, R, S = cfg
    spec = ops.GridSpec(L, F, lg, 16, 2048)
    o, d, area, s, e, eu = _sample_rays(R, S, seed=7)
    do, dd, da, edges = dev(o), dev(d), dev(area), dev(eu)
    st, en = edges[:, :-1], edges[:, 1:]
    g = synth.normal((R, S, L * F), 13)
    cut = (synth.uniform((R,), 0, 1, 14) * S).astype(np.int64)  # the ray is opaque from sample cut[r] on
    g[np.arange(S)[None, :] >= cut[:, None]] = 0.0
    g[synth.uniform((R, S), 0, 1, 15) < 0.1] = 0.0  # scattered silent samples inside runs of equal cells
    g[::7] = 0.0  # whole rays
    go = dev(g.reshape(R * S, L * F))
    assert 0.5 < float((go.abs().amax(-1) == 0).float().mean()) < 0.9
    monkeypatch.setattr(ops, "_FORCE_ATOMIC_SCATTER", False)
    monkeypatch.setattr(ops, "_BINNED_MIN_SAMPLES", 1)
    binned = ops.encode_bwd(spec, 1.0, do, dd, da, st, en, go)
    neg = torch.where(go == 0, -torch.zeros_like(go), go)
    assert torch.equal(ops.encode_bwd(spec, 1.0, do, dd, da, st, en, neg), binned)
    monkeypatch.setattr(ops, "_FORCE_ATOMIC_SCATTER", True)
    atomic = ops.encode_bwd(spec, 1.0, do, dd, da, st, en, go)
    assert rel_l2(host(binned), host(atomic)) < 2e-6
    assert (binned - atomic).abs().max() <= 1e-5 * atomic.abs().max()
    # an all-zero gradient gives an all-zero table gradient (every slice is "empty" and zero-filled)
    monkeypatch.setattr(ops, "_FORCE_ATOMIC_SCATTER", False)
    assert float(ops.encode_bwd(spec, 1.0, do, dd, da, st, en, torch.zeros_like(go)).abs().max()) == 0.0
    if F == 1:  # the proposal density backward flags by its own incoming gradient
        ps = ops.ProposalSpec(spec, dev(synth.hash_table(L << lg, 1, seed=3, scale=0.5)), 1.0, dev(synth.normal((1, L), 5)))
        dens = ops.proposal_density_fwd(ps, do, dd, da, st, en)
        gd = dev(g[..., 0].copy())
        gt, gdec = ops.proposal_density_bwd(ps, do, dd, da, st, en, dens, gd)
        monkeypatch.setattr(ops, "_FORCE_ATOMIC_SCATTER", True)
        gt2, gdec2 = ops.proposal_density_bwd(ps, do, dd, da, st, en, dens, gd)
        assert rel_l2(host(gt), host(gt2)) < 2e-6 and rel_l2(host(gdec), host(gdec2)) < 1e-5


@pytest.mark.parametrize("dims", [(32, 2, 64, 33), (32, 2, 32, 33), (48, 3, 64, 32), (48, 3, 32, 32), (64, 3, 64, 32),
                                  (64, 3, 32, 32)])
def test_mlp_register_chained_shapes_vs_oracle(ops, dims):
    """F2 fast path (mlp_chain.hip): NeuRAD's own MLP shapes, forward + data/weight gradients against the oracle on
    ragged batch sizes (partial 16-sample tiles, one tile, many workgroups), with and without biases."""
    i, n, w, o = dims
    dd = [i] + [w] * (n - 1) + [o]
    for use_bias in (True, False):
        ws, bs = [], []
        for k in range(n):
            wk, bk = synth.linear(dd[k + 1], dd[k], 900 + k)
            ws.append(wk), bs.append(bk if use_bias else None)
        dws, dbs = [dev(a) for a in ws], [None if b is None else dev(b) for b in bs]
        for N in (1, 15, 16, 17, 1000, 70001):
            x = synth.normal((N, i), seed=N)
            go = synth.normal((N, o), seed=N + 1)
            if N >= 1000:  # runs of exactly-zero gradient rows (samples behind a surface): whole 16-row tiles are
                go[200:488] = 0.0  # skipped by the kernels, partially zero ones are not -- results must not change
                go[N - 100:] = 0.0
                go[3::5] = 0.0
            y, hidden = ops.mlp_fwd(dev(x), dws, dbs, save_hidden=True)
            ref_y, acts = O.mlp_fwd(x, ws, bs, return_hidden=True)
            assert rel_l2(host(y), ref_y) < TIGHT, (dims, N)
            gx, gws, gbs = ops.mlp_bwd(dev(x), hidden, dev(go), dws, dbs)
            # the ReLU masks come from the activations the device saved: with 9 M hidden units a pre-activation
            # within one rounding of zero flips its mask between summation orders, which is not a backward error
            hh = host(hidden)
            for k in range(n - 1):
                assert rel_l2(hh[:, k * w:(k + 1) * w], acts[k + 1]) < TIGHT, (dims, N, k)
            acts_dev = [x] + [hh[:, k * w:(k + 1) * w] for k in range(n - 1)] + [host(y)]
            rx, rws, rbs = O.mlp_bwd(acts_dev, ws, go)
            assert rel_l2(host(gx), rx) < TIGHT, (dims, N)
            for k in range(n):
                assert rel_l2(host(gws[k]), rws[k]) < TIGHT, (dims, N, k)
                if use_bias:
                    assert rel_l2(host(gbs[k]), rbs[k]) < TIGHT, (dims, N, k)


@pytest.mark.parametrize("width", [32, 64])
def test_field_feature_head_backward_one_pass_vs_composed(ops, width):
    """nrhip_field_feature_bwd == nrhip_mlp_bwd + the residual add + column 0 (the composition it replaces), and the
    oracle's MLP backward, on ragged batch sizes (partial tiles, a single row, many workgroups)"""
    ws, bs = [], []
    dd = [48, width, width, 32]
    for k in range(3):
        wk, bk = synth.linear(dd[k + 1], dd[k], 700 + k)
        ws.append(wk), bs.append(bk)
    dws, dbs = [dev(a) for a in ws], [dev(b) for b in bs]
    assert ops.field_feature_bwd_supported(dws, dbs)
    for N in (1, 15, 16, 17, 1000, 70001):
        x = synth.normal((N, 48), seed=N)
        gf = synth.normal((N, 32), seed=N + 1)
        g0 = synth.normal((N,), seed=N + 2)
        if N >= 1000:
            gf[200:488] = 0.0
            gf[3::5] = 0.0
        y, hidden = ops.mlp_fwd(dev(x), dws, dbs, save_hidden=True)
        g_geo, gws, gbs = ops.field_feature_bwd(dev(x), hidden, dev(gf), dev(g0), dws, dbs)
        gx, rws, rbs = ops.mlp_bwd(dev(x), hidden, dev(gf), dws, dbs)
        ref = np.concatenate([g0[:, None], gf + host(gx)[:, :32]], 1)
        assert g_geo.shape == (N, 33)
        np.testing.assert_array_equal(host(g_geo)[:, 0], g0)
        assert rel_l2(host(g_geo), ref) < 1e-6, (width, N)
        for k in range(3):  # same kernel body for the weight gradients: identical sums
            assert rel_l2(host(gws[k]), host(rws[k])) < 1e-6 and rel_l2(host(gbs[k]), host(rbs[k])) < 1e-6, (width, N, k)
        hh = host(hidden)
        acts_dev = [x] + [hh[:, k * width:(k + 1) * width] for k in range(2)] + [host(y)]
        ox, ows, _ = O.mlp_bwd(acts_dev, ws, gf)
        assert rel_l2(host(g_geo)[:, 1:], gf + ox[:, :32]) < TIGHT and rel_l2(host(gws[0]), ows[0]) < TIGHT, (width, N)


def test_encode_bwd_binned_degenerate_distribution(ops, monkeypatch):
    """Every ray identical (one line of cells), half of the samples at one point: a handful of table entries receive
    almost all records.  The radix partition sizes its queues exactly, so this is slow-ish but exact."""
    L, F, lg, R, S = 8, 4, 18, 2048, 64
    spec = ops.GridSpec(L, F, lg, 32, 4096)
    o, d, area, s, e, eu = _sample_rays(1, S, seed=9)
    do, dd, da = dev(np.repeat(o, R, 0)), dev(np.repeat(d, R, 0)), dev(np.repeat(area, R, 0))
    edges = dev(np.repeat(eu, R, 0)).clone()
    edges[: R // 2] = torch.linspace(5.0, 5.0001, S + 1, device="cuda")  # half of the rays: all samples in one cell
    go = dev(synth.normal((R * S, L * F), 3))
    st, en = edges[:, :-1], edges[:, 1:]
    monkeypatch.setattr(ops, "_BINNED_MIN_SAMPLES", 1)
    monkeypatch.setattr(ops, "_FORCE_ATOMIC_SCATTER", False)
    binned = ops.encode_bwd(spec, 1.0, do, dd, da, st, en, go)
    monkeypatch.setattr(ops, "_FORCE_ATOMIC_SCATTER", True)
    atomic = ops.encode_bwd(spec, 1.0, do, dd, da, st, en, go)
    assert torch.isfinite(binned).all() and (binned != 0).sum() < 20000
    assert rel_l2(host(binned), host(atomic)) < 1e-5  # the fp32 atomics lose bits summing 1e5 terms into one entry


def test_sampler_losses_vs_reference_autograd(ops):
    """zipnerf_interlevel_loss / distortion_loss -- one wavefront per ray against the reference's
    own values and autograd gradients (tests/golden/losses.npz), the oracle on ragged sizes, and autograd plumbing."""
    from types import SimpleNamespace

    from neurad_studio_amd.model_components import losses as hl

    g = load_golden("losses")
    R = g["w0"].shape[0]
    total = 0.0
    for cp, wp, gw, r in [(g["sd0"], g["w0"], g["g_w0"], 0.03), (g["sd1"], g["w1"], g["g_w1"], 0.003)]:
        loss, grad = ops.interlevel_loss_level(dev(g["sdf"]), dev(g["wf"]), dev(cp), dev(wp), r)
        ref_loss, _, ref_grad = O.interlevel_loss_level(g["sdf"], g["wf"], cp, wp, r)
        assert rel_l2(host(loss), ref_loss) < TIGHT
        assert rel_l2(host(grad) / R, gw) < TOL and rel_l2(host(grad), ref_grad) < TIGHT
        total += host(loss).mean()
    assert abs(total - g["interlevel"]) / g["interlevel"] < TIGHT
    dl, dg = ops.distortion_loss_rays(dev(g["sdf"]), dev(g["wf"]))
    assert abs(host(dl).mean() - g["distortion"]) / g["distortion"] < TIGHT and rel_l2(host(dg) / R, g["g_wf"]) < TIGHT

    # the reference-shaped functions, through autograd
    def samples(sd):
        t = dev(sd)
        return SimpleNamespace(spacing_starts=t[:, :-1, None], spacing_ends=t[:, 1:, None])

    w0, w1, wf = (dev(g[k]).requires_grad_(True) for k in ("w0", "w1", "wf"))
    rsl = [samples(g["sd0"]), samples(g["sd1"]), samples(g["sdf"])]
    wl = [w0[..., None], w1[..., None], wf[..., None]]
    il = hl.zipnerf_interlevel_loss(wl, rsl)
    (il * 0.5).backward()
    assert abs(il.item() - g["interlevel"]) / g["interlevel"] < TIGHT
    assert rel_l2(host(w0.grad), 0.5 * g["g_w0"]) < TOL and rel_l2(host(w1.grad), 0.5 * g["g_w1"]) < TOL and wf.grad is None
    dist = hl.distortion_loss(wl, rsl)
    dist.backward()
    assert abs(dist.item() - g["distortion"]) / g["distortion"] < TIGHT and rel_l2(host(wf.grad), g["g_wf"]) < TIGHT

    # ragged sizes / degenerate bins against the oracle: zero-width bins, zero weights, one fine sample
    for (R2, sf, sp) in [(3, 1, 5), (5, 7, 130), (2, 128, 512), (4, 33, 64)]:
        c = np.sort(synth.uniform((R2, sf + 1), 0.0, 1.0, seed=sf), -1).astype(np.float32)
        c[:, 0], c[:, -1] = 0.0, 1.0
        cp = np.sort(synth.uniform((R2, sp + 1), 0.0, 1.0, seed=sp), -1).astype(np.float32)
        cp[:, 0], cp[:, -1] = 0.0, 1.0
        if sp > 8:
            cp[:, 3] = cp[:, 4]  # a zero-width proposal bin
        w = synth.uniform((R2, sf), 0.0, 1.0, seed=7).astype(np.float32)
        w = w / w.sum(-1, keepdims=True) * 0.8
        wp = synth.uniform((R2, sp), 0.0, 2.0 / sp, seed=8).astype(np.float32)
        wp[:, 0] = 0.0
        loss, grad = ops.interlevel_loss_level(dev(c), dev(w), dev(cp), dev(wp), 0.03)
        rl, _, rg = O.interlevel_loss_level(c, w, cp, wp, 0.03)
        assert rel_l2(host(loss), rl) < TOL and rel_l2(host(grad), rg) < TOL, (R2, sf, sp)
        dl, dg = ops.distortion_loss_rays(dev(cp), dev(wp))
        rdl, rdg = O.distortion_loss_rays(cp, wp)
        assert rel_l2(host(dl), rdl) < TIGHT and rel_l2(host(dg), rdg) < TIGHT


def test_power_sampler_sky_stretch_folded(ops):
    """last_edge > 0 == the model's `frustums.ends[:, -1] = sky_distance` applied to the sampler's euclidean bins."""
    fars = dev(synth.uniform((33,), 10.0, 20000.0, seed=4))
    sp0, eu0 = ops.power_sampler(None, fars, 37)
    sp1, eu1 = ops.power_sampler(None, fars, 37, last_edge=20000.0)
    eu0[:, -1] = 20000.0
    assert torch.equal(sp0, sp1) and torch.equal(eu0, eu1)


@pytest.mark.parametrize("shape", [(301, 32, 32), (17, 129, 3), (64, 31, 48), (5, 1, 1)])
def test_accumulate_along_rays_backward_kernel_vs_torch(ops, shape):
    """nrhip_accumulate_along_rays_bwd == autograd of sum_s w[r,s] v[r,s,c] (torch fp32), flat-walk and generic channel
    counts, with either gradient switched off."""
    R, S, Cc = shape
    w, v, g = dev(synth.uniform((R, S), 0, 1, 1)), dev(synth.normal((R, S, Cc), 2)), dev(synth.normal((R, Cc), 3))
    wt, vt = w.clone().requires_grad_(True), v.clone().requires_grad_(True)
    (wt[..., None] * vt).sum(1).backward(g)
    gw, gv = ops.accumulate_along_rays_bwd(w, v, g)
    assert rel_l2(host(gw), host(wt.grad)) < 1e-6 and rel_l2(host(gv), host(vt.grad)) < 1e-6
    assert ops.accumulate_along_rays_bwd(w, v, g, need_grad_values=False)[1] is None
    assert torch.equal(ops.accumulate_along_rays_bwd(w, v, g, need_grad_weights=False)[1], gv)


@pytest.mark.parametrize("cfg", [(40000, 42, 16, True), (1000, 7, 16, False), (5000, 3000, 8, True)])
def test_embedding_lerp_kernels_vs_torch_embedding(cfg):
    """C3 (models/neurad.py:423-441): e_lo (1 - frac) + e_hi frac, forward bit for bit the torch expression, table gradient
    against nn.Embedding's autograd -- LDS-image path (small tables) and the global-atomic path (3000 x 8 > 32 KB)."""
    from neurad_studio_amd import autograd as ag

    R, E, D, temporal = cfg
    torch.manual_seed(4)
    emb = torch.nn.Embedding(E, D).cuda()
    lo = torch.randint(0, E, (R,), device="cuda")
    hi = (lo + 1).clamp_max(E - 1)
    frac = torch.rand(R, device="cuda")
    g = torch.randn(R, D, device="cuda")
    ref = emb(lo) * (1 - frac[:, None]) + emb(hi) * frac[:, None] if temporal else emb(lo)
    ref.backward(g)
    want = emb.weight.grad.clone()
    emb.weight.grad = None
    out = ag.EmbeddingLerpFn.apply(emb.weight, lo, hi if temporal else None, frac if temporal else None)
    assert torch.equal(out, ref.detach())
    out.backward(g)
    assert rel_l2(host(emb.weight.grad), host(want)) < 2e-6


def test_encode_bwd_binned_writes_the_fp16_gradient_of_an_fp16_storage_table(ops, monkeypatch):
    """round 5: with out_dtype = fp16 the partition's `reduce` writes the gradient in the table's own storage type (what
    autograd wants for an fp16-storage table) instead of an fp32 image + a cast pass: every element equals the fp16 rounding
    of the fp32 result -- identical integer accumulation, one rounding at the store"""
    L, F, lg, R, S = 8, 4, 15, 1100, 32
    spec = ops.GridSpec(L, F, lg, 16, 2048)
    o, d, area, s, e, eu = _sample_rays(R, S, seed=9)
    do, dd, da, edges = dev(o), dev(d), dev(area), dev(eu)
    go = dev(synth.normal((R * S, L * F), 19))
    go[::5] = 0  # silent samples; some slices stay empty at this size
    st, en = edges[:, :-1], edges[:, 1:]
    monkeypatch.setattr(ops, "_FORCE_ATOMIC_SCATTER", False)
    monkeypatch.setattr(ops, "_BINNED_MIN_SAMPLES", 1)
    g32 = ops.encode_bwd(spec, 1.0, do, dd, da, st, en, go)
    g16 = ops.encode_bwd(spec, 1.0, do, dd, da, st, en, go, out_dtype=torch.float16)
    assert g32.dtype == torch.float32 and g16.dtype == torch.float16 and g16.shape == g32.shape
    assert torch.equal(g16, g32.half())
    bad = go.clone()
    bad[3, 1] = float("nan")
    assert not torch.isfinite(ops.encode_bwd(spec, 1.0, do, dd, da, st, en, bad, out_dtype=torch.float16).float()).all()


def test_binned_table_gradient_is_fp32_whatever_the_tables_storage_type(ops, monkeypatch):
    """ABI 510 (include/neurad_hip.h): nrhip_encode_bwd_binned / nrhip_hashgrid_bwd_binned ignore g->param_dtype -- a C caller
    that passes the descriptor of its fp16-storage TABLE with an fp32 grad_table gets an fp32 gradient (ABI 500 wrote fp16
    halves into it); the fp16 form is the explicit nrhip_encode_bwd_binned_f16"""
    import ctypes as C

    from neurad_studio_amd import _lib

    L, F, lg, R, S = 8, 4, 14, 300, 32
    spec = ops.GridSpec(L, F, lg, 16, 2048)
    o, d, area, s, e, eu = _sample_rays(R, S, seed=4)
    do, dd, da, edges = dev(o), dev(d), dev(area), dev(eu)
    go = dev(synth.normal((R * S, L * F), 23))
    st, en = edges[:, :-1], edges[:, 1:]
    monkeypatch.setattr(ops, "_FORCE_ATOMIC_SCATTER", False)
    monkeypatch.setattr(ops, "_BINNED_MIN_SAMPLES", 1)
    want = ops.encode_bwd(spec, 1.0, do, dd, da, st, en, go)
    table16 = torch.zeros((spec.table_rows, F), device="cuda", dtype=torch.float16)
    g = spec.c_grid(table16)  # param_dtype = 1: the TABLE is fp16 storage
    assert g.param_dtype == 1
    r, keep = ops._c_rays(do, dd, da, st, en)
    ws = ops._table_grad_workspace(g, R * S, do.device)
    gt = torch.full((spec.table_rows, F), float("nan"), device="cuda", dtype=torch.float32)
    _lib.call("nrhip_encode_bwd_binned", C.byref(g), 1.0, C.byref(r), ops._ptr(go), ops._ptr(gt), 1, ops._ptr(ws), ws.numel(),
              ops._stream())
    assert torch.equal(gt, want)
    x = dev(synth.uniform((500, 3), 0, 1, 5))
    g2 = dev(synth.normal((500, L * F), 6))
    want2 = ops.hashgrid_bwd(spec, None, x, g2)
    gt2 = torch.full_like(gt, float("nan"))
    ws2 = ops._table_grad_workspace(g, 500, x.device)
    _lib.call("nrhip_hashgrid_bwd_binned", C.byref(g), ops._ptr(x), ops._ptr(g2), 500, ops._ptr(gt2), 1, ops._ptr(ws2),
              ws2.numel(), ops._stream())
    assert torch.equal(gt2, want2)
