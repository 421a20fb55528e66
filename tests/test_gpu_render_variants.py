"""The fused render kernel's scheduling features against the CPU oracle: the software pipeline across ray boundaries,
processing orders (`nrhip_rays.order` from nrhip_ray_order -- a locality hint that must never change a result) and the
early-ray-termination option: exact when off, bounded by `early_stop_eps` when on."""
import numpy as np
import pytest
import torch

import neurad_oracle as O
from conftest import rel_l2
from test_gpu_parity import RENDER_CFGS, TOL, _sample_rays, dev, field_params, host, to_spec

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from neurad_studio_amd import ops as _ops

    return _ops


def test_ray_order_is_a_permutation_grouped_by_region(ops):
    R = 5000
    g = torch.Generator(device="cuda").manual_seed(5)
    o = torch.randn((R, 3), device="cuda", generator=g) * 5
    d = torch.randn((R, 3), device="cuda", generator=g)
    d = d / d.norm(dim=-1, keepdim=True)
    order = ops.ray_order(o, d, static_scale=100.0, t_ref=20.0)
    assert order.dtype == torch.int32 and order.shape == (R,)
    assert torch.equal(torch.sort(order.long()).values, torch.arange(R, device="cuda"))
    # neighbours in the order look at nearby points: mean distance of consecutive key points far below a random order's
    p = (o + 20.0 * d)[order.long()]
    near = (p[1:] - p[:-1]).norm(dim=-1).mean()
    rand = ((o + 20.0 * d)[1:] - (o + 20.0 * d)[:-1]).norm(dim=-1).mean()
    assert float(near) < 0.5 * float(rand)
    assert ops.ray_order(o[:0], d[:0], 100.0).shape == (0,)
    nan = o.clone()
    nan[3] = float("nan")  # garbage rays must still yield a valid permutation
    assert torch.equal(torch.sort(ops.ray_order(nan, d, 100.0).long()).values, torch.arange(R, device="cuda"))


@pytest.mark.parametrize("key_bits", [0, 5])
def test_ray_order_many_workgroups_same_buckets_as_one_workgroup(ops, key_bits, monkeypatch):
    """round 5: above 16 384 rays the counting sort runs over many workgroups (nrhip_ray_order_large: keys + global histogram,
    bucket scan, placement).  Same keys, same buckets in the same sequence (the order inside a bucket is unspecified in
    both passes): a valid permutation, NaN rays included, whose bucket boundaries are the single-workgroup pass's"""
    R = 70001
    g = torch.Generator(device="cuda").manual_seed(7)
    o = torch.randn((R, 3), device="cuda", generator=g) * torch.tensor([20.0, 20.0, 0.3], device="cuda")
    d = torch.randn((R, 3), device="cuda", generator=g)
    d = d / d.norm(dim=-1, keepdim=True)
    o[11] = float("nan")
    large = ops.ray_order(o, d, static_scale=100.0, t_ref=30.0, key_bits=key_bits)
    monkeypatch.setattr(ops, "_RAY_ORDER_LARGE", 1 << 40)
    small = ops.ray_order(o, d, static_scale=100.0, t_ref=30.0, key_bits=key_bits)
    ar = torch.arange(R, device="cuda")
    for od in (large, small):
        assert od.dtype == torch.int32 and torch.equal(torch.sort(od.long()).values, ar)

    def cuts(a, b):
        """walk the rays in order ``a``; q = their positions in order ``b``.  A prefix of the walk is a union of whole buckets
        of ``b`` exactly where it is {0..i} as a set, i.e. where its running maximum equals i"""
        pos_b = torch.empty(R, dtype=torch.long, device="cuda")
        pos_b[b.long()] = ar
        q = pos_b[a.long()]
        return torch.cummax(q, 0).values == ar

    c_ls, c_sl = cuts(large, small), cuts(small, large)
    # (prefix-set equality is symmetric, so the two cut sets coincide by construction; what tests the passes is that there
    #  are MANY cuts -- one at least per non-empty bucket -- which only holds if both put the same rays into the same buckets
    #  in the same bucket sequence)
    assert torch.equal(c_ls, c_sl) and int(c_ls.sum()) > (300 if key_bits == 5 else 40), int(c_ls.sum())


def test_ray_order_five_bit_keys(ops):
    """key_bits = 5: 32 768 buckets (128 KB LDS histogram), both entry points"""
    R = 6000
    g = torch.Generator(device="cuda").manual_seed(6)
    o = torch.randn((R, 3), device="cuda", generator=g) * 5
    d = torch.randn((R, 3), device="cuda", generator=g)
    d = d / d.norm(dim=-1, keepdim=True)
    key_pts = o + 20.0 * d
    spread = lambda od: float((key_pts[od.long()][1:] - key_pts[od.long()][:-1]).norm(dim=-1).mean())  # noqa: E731
    o4 = ops.ray_order(o, d, static_scale=100.0, t_ref=20.0)
    o5 = ops.ray_order(o, d, static_scale=100.0, t_ref=20.0, key_bits=5)
    fars = torch.full((R,), 1000.0, device="cuda")
    sp, eu, o5f = ops.power_sampler_ordered(None, fars, 16, o, d, 100.0, t_ref=20.0, key_bits=5)
    for od in (o5, o5f):
        assert torch.equal(torch.sort(od.long()).values, torch.arange(R, device="cuda"))
        assert spread(od) <= spread(o4) * 1.05  # finer cells group at least as tightly
    sp0, eu0 = ops.power_sampler(None, fars, 16)
    assert torch.equal(sp, sp0) and torch.equal(eu, eu0)


def test_power_sampler_and_ordering_pass_in_one_launch(ops):
    """nrhip_power_sampler_ordered == nrhip_power_sampler + nrhip_ray_order: identical bins (eval and injected jitter), and an
    order that is a permutation with the same grouping quality (the order inside a bucket is unspecified in both)"""
    R, S = 4099, 128
    g = torch.Generator(device="cuda").manual_seed(8)
    o = torch.randn((R, 3), device="cuda", generator=g) * 5
    d = torch.randn((R, 3), device="cuda", generator=g)
    d = d / d.norm(dim=-1, keepdim=True)
    fars = torch.rand((R,), device="cuda", generator=g) * 1000 + 50
    nears = torch.rand((R,), device="cuda", generator=g)
    for t_rand in (None, torch.rand((R, S + 1), device="cuda", generator=g)):
        sp0, eu0 = ops.power_sampler(nears, fars, S, t_rand=t_rand, last_edge=20000.0)
        sp1, eu1, order = ops.power_sampler_ordered(nears, fars, S, o, d, 100.0, t_rand=t_rand, last_edge=20000.0, t_ref=20.0)
        assert torch.equal(sp0, sp1) and torch.equal(eu0, eu1)
        assert order.dtype == torch.int32 and torch.equal(torch.sort(order.long()).values, torch.arange(R, device="cuda"))
        sep = ops.ray_order(o, d, static_scale=100.0, t_ref=20.0)
        key_pts = o + 20.0 * d
        spread = lambda od: float((key_pts[od.long()][1:] - key_pts[od.long()][:-1]).norm(dim=-1).mean())  # noqa: E731
        assert abs(spread(order) / spread(sep) - 1) < 0.05
    for S2 in (1, 7, 1500):  # a workgroup's 1024 edges span 512 rays / 128 rays / less than one ray
        sp0, eu0 = ops.power_sampler(nears, fars, S2, last_edge=20000.0)
        sp1, eu1, _ = ops.power_sampler_ordered(nears, fars, S2, o, d, 100.0, last_edge=20000.0)
        assert torch.equal(sp0, sp1) and torch.equal(eu0, eu1), S2
    sp, eu, order = ops.power_sampler_ordered(None, fars[:3], 7, o[:3], d[:3], 100.0)  # fewer rays than one workgroup
    assert sp.shape == (3, 8) and sorted(order.tolist()) == [0, 1, 2]
    assert ops.power_sampler_ordered(None, fars[:0], 7, o[:0], d[:0], 100.0)[2].shape == (0,)


@pytest.mark.parametrize("ordered", [False, True])
@pytest.mark.parametrize("cfg", RENDER_CFGS)
def test_render_with_processing_order_vs_oracle(ops, cfg, ordered):
    L, F, lg, mn, mx, H, use_sdf, R, S = cfg
    p = field_params(use_sdf=use_sdf, L=L, F=F, lg=lg, H=H, mn=mn, mx=mx, scale=2.0 if use_sdf else 0.5)
    if use_sdf:
        p.beta = 3.0
    fs = to_spec(ops, p)
    o, d, area, s, e, eu = _sample_rays(R, S, seed=R + S)
    ref = O.render_rays(p, o, d, area, s, e)
    edges = dev(eu)
    order = ops.ray_order(dev(o), dev(d), p.static_scale) if ordered else None
    feats, depth, acc, w = ops.render_fwd(fs, dev(o), dev(d), dev(area), edges[:, :-1], edges[:, 1:],
                                          return_weights=True, order=order)
    assert rel_l2(host(w), ref["weights"]) < TOL
    assert rel_l2(host(feats), ref["features"]) < TOL
    assert rel_l2(host(depth), ref["depth"]) < TOL or np.abs(host(depth) - ref["depth"]).max() < 1e-5
    assert rel_l2(host(acc), ref["accumulation"]) < TOL


def test_order_never_changes_results_more_rays_than_waves(ops):
    """R far above the persistent grid's wave count: every wave walks several rays, the flattened (ray, tile) pipeline
    crosses ray boundaries, ragged S.  Any processing order gives bit-identical per-ray results (each ray is computed by
    one wave with the same arithmetic), for the composited and the per-sample / training entry points."""
    p = field_params(use_sdf=True, L=16, F=2, lg=14, H=64, mn=16, mx=1024, scale=1.0)
    p.beta = 2.0
    fs = to_spec(ops, p)
    R, S = 9000, 37
    o, d, area, s, e, eu = _sample_rays(R, S, seed=3)
    edges, do, dd, da = dev(eu), dev(o), dev(d), dev(area)
    orders = [None, ops.ray_order(do, dd, p.static_scale), torch.randperm(R, device="cuda").to(torch.int32),
              torch.arange(R - 1, -1, -1, device="cuda", dtype=torch.int32)]
    base = ops.render_fwd(fs, do, dd, da, edges[:, :-1], edges[:, 1:], return_weights=True)
    fbase = ops.field_fwd(fs, do, dd, da, edges[:, :-1], edges[:, 1:])
    tbase = ops.field_fwd_train(fs, do, dd, da, edges[:, :-1], edges[:, 1:])
    for od in orders[1:]:
        for a, b in zip(ops.render_fwd(fs, do, dd, da, edges[:, :-1], edges[:, 1:], return_weights=True, order=od), base):
            assert torch.equal(a, b)
        for a, b in zip(ops.field_fwd(fs, do, dd, da, edges[:, :-1], edges[:, 1:], order=od), fbase):
            assert torch.equal(a, b)
        got = ops.field_fwd_train(fs, do, dd, da, edges[:, :-1], edges[:, 1:], order=od)
        for a, b in zip(got[0] + got[1], tbase[0] + tbase[1]):
            assert torch.equal(a, b)
    sl = slice(8990, 9000)  # the tail rays (last pipeline stages) against the oracle
    ref = O.render_rays(p, o[sl], d[sl], area[sl], s[sl], e[sl])
    assert rel_l2(host(base[0][sl]), ref["features"]) < TOL
    with pytest.raises(ValueError):
        ops.render_fwd(fs, do, dd, da, edges[:, :-1], edges[:, 1:], order=orders[1][:-1])


@pytest.mark.parametrize("use_sdf", [True, False])
def test_early_ray_termination_bounded(ops, use_sdf):
    """early_stop_eps: rays stop once their transmittance is below eps.  What is dropped weighs < eps in total, so
    accumulation changes by < eps, features by < eps * max|feature|; weights of skipped samples come back as zeros and
    everything before the cut is bit-identical.  eps = 0 is the exact path."""
    p = field_params(use_sdf=use_sdf, L=8, F=4, lg=11, H=32, scale=2.0 if use_sdf else 0.5)
    if use_sdf:
        p.beta = 6.0  # alphas around 0.5: transmittance falls below 1e-3 after a dozen samples
    R, S = 300, 96
    eps = 1e-3
    o, d, area, s, e, eu = _sample_rays(R, S, seed=21)
    # density head: shift the logit until the medium is opaque enough to cut rays but not so dense that the exact
    # weights underflow to 0 on their own
    for bump in (0.0, -3.0, -2.0, -1.0, 1.0, 2.0, 3.0):
        if not use_sdf:
            p.geo_b[1][0] += bump
        fs = to_spec(ops, p)
        args = (fs, dev(o), dev(d), dev(area), dev(s), dev(e))
        f0, d0, a0, w0 = ops.render_fwd(*args, return_weights=True)
        # the kernel's rule: the tile whose ENTERING transmittance is below eps is the last one evaluated
        T_enter = 1.0 - torch.cumsum(w0.double(), -1)[:, 15::16]          # behind tiles 0, 1, ...
        T_enter = torch.cat([torch.ones_like(T_enter[:, :1]), T_enter[:, :-1]], 1)
        stop = T_enter < eps * 0.5                                         # margin: fp32 carry vs this fp64 sum
        last = torch.where(stop.any(1), stop.float().argmax(1), torch.full((R,), S // 16 - 1, device="cuda"))
        skipped = torch.arange(S, device="cuda")[None, :] >= (16 * (last + 1))[:, None]
        if use_sdf or (skipped.float().mean() > 0.25 and bool((w0[skipped] > 0).float().mean() > 0.5)):
            break
        if not use_sdf:
            p.geo_b[1][0] -= bump
    assert skipped.float().mean() > 0.25, "the test scene must actually terminate rays early"
    f1, d1, a1, w1 = ops.render_fwd(*args, return_weights=True, early_stop_eps=eps)
    assert bool((w1[skipped] == 0).all())
    kept = ~((w1 == 0) & (w0 != 0))
    assert torch.equal(w1[kept], w0[kept])
    assert float((w0 * (~kept)).sum(-1).max()) <= eps * 1.01  # per ray: what was skipped weighs < eps
    assert float((a1 - a0).abs().max()) <= eps * 1.01
    fmax = float(f0.abs().max()) + 1.0
    assert float((f1 - f0).abs().max()) <= 3 * eps * fmax
    # the sky residual (1 - acc on the last sample) is dropped with the tail: it is < eps on a terminated ray
    assert float((d1 - d0).abs().max()) <= eps * float(e.max())
    # a threshold nothing reaches: same arithmetic, same result
    f2, d2, a2, w2 = ops.render_fwd(*args, return_weights=True, early_stop_eps=1e-30)
    assert torch.equal(f2, f0) and torch.equal(w2, w0) and torch.equal(a2, a0)
    from neurad_studio_amd._lib import NeuradHipError

    with pytest.raises(NeuradHipError):
        ops.render_fwd(*args, early_stop_eps=1.5)


def test_split_bf16_matrix_products_are_fp32_equivalent(switches):
    """NRHIP_MLP_SPLIT_BF16: the fused render kernel's MLP layers as 3-way split bf16 on the matrix cores (six bf16 MFMAs
    per 16 x 16 block keep every product term above 2^-24).  Same outputs as the fp32-MFMA kernel to fp32 rounding, and the
    same parity against the oracle -- on config[1]'s shape (16 levels, 64-wide) and on 8 x 4 levels."""
    from neurad_studio_amd import ops

    for L, F, mn, mx in ((16, 2, 16, 1024), (8, 4, 32, 8192)):
        p = field_params(L=L, F=F, lg=12, H=64, mn=mn, mx=mx)
        fs = to_spec(ops, p)
        R, S = 300, 72
        o, d, area, s, e, _ = _sample_rays(R, S, seed=11)
        args = (fs, dev(o), dev(d), dev(area), dev(s), dev(e))
        switches.unset("NRHIP_MLP_SPLIT_BF16")
        switches.set("NRHIP_MLP_PAIRS", "0")  # (the default since round 5 is the fp16-pair form, tested below)
        f32 = ops.render_fwd(*args, return_weights=True)
        switches.set("NRHIP_MLP_SPLIT_BF16", "1")
        spl = ops.render_fwd(*args, return_weights=True)
        for a, b in zip(f32, spl):
            assert rel_l2(host(b), host(a)) < 1e-6
        assert not torch.equal(spl[0], f32[0])  # (a different kernel did run)
        ref = O.render_rays(p, o[:32], d[:32], area[:32], s[:32], e[:32])
        assert rel_l2(host(spl[0][:32]), ref["features"]) < 1e-5


def test_fp16_pair_matrix_products_are_fp32_equivalent(switches):
    """The default of the composited kernels since round 5 (NRHIP_MLP_PAIRS=0 switches it off): the fused render
    kernel's MLP layers as fp16 pairs on the matrix cores (x = fp16(x) + fp16(x - fp16(x)),
    three v_mfma_f32_16x16x32_f16 per 32 inputs; the tile runs in units of 2^6 and the weights are staged x 2^7 so that the
    pairs keep 22+ bits where the network lives).  Same outputs as the fp32-MFMA kernel to fp32 rounding and the same parity
    against the oracle on all three 64-wide grid shapes, fp32 and fp16-storage tables; inputs that do not fit an fp16 pair
    (activations beyond 1000, a weight beyond 500) take the fp32 products tile by tile and give the default kernel's numbers;
    an untrained field (table entries of 1e-4, the Instant-NGP initialisation) stays within 2e-6."""
    from neurad_studio_amd import ops

    R, S = 300, 72
    o, d, area, s, e, _ = _sample_rays(R, S, seed=11)
    rays = (dev(o), dev(d), dev(area), dev(s), dev(e))

    def both(fs):
        switches.set("NRHIP_MLP_PAIRS", "0")
        a = ops.render_fwd(fs, *rays, return_weights=True)
        switches.set("NRHIP_MLP_PAIRS", "1")
        b = ops.render_fwd(fs, *rays, return_weights=True)
        switches.unset("NRHIP_MLP_PAIRS")
        return a, b

    for L, F, mn, mx, H in ((16, 2, 16, 1024, 64), (8, 4, 32, 8192, 64), (4, 8, 32, 2048, 64), (8, 4, 32, 8192, 32),
                            (16, 2, 16, 1024, 32), (4, 8, 32, 2048, 32)):
        p = field_params(L=L, F=F, lg=12, H=H, mn=mn, mx=mx)
        ref = O.render_rays(p, o[:32], d[:32], area[:32], s[:32], e[:32])
        for half in (False, True):
            f32, prs = both(to_spec(ops, p, half=half))
            for a, b in zip(f32, prs):
                assert rel_l2(host(b), host(a)) < 1e-6, (L, F, H, half)
            assert not torch.equal(prs[0], f32[0])  # (a different kernel did run)
            if not half:
                assert rel_l2(host(prs[0][:32]), ref["features"]) < 1e-5
                assert rel_l2(host(prs[1][:32]), ref["depth"]) < 1e-5
    # --- the exits of the fast path, at both widths
    for H in (64, 32):
        # (1) activations beyond the pair's range in SOME tiles: half of the table's rows scaled up
        big = field_params(L=16, F=2, lg=12, H=H, mn=16, mx=1024)
        big.grid.table[::2] *= 3.0e3
        f32, prs = both(to_spec(ops, big))
        for a, b in zip(f32, prs):  # (SDF values of 1e4 and more: the compositing amplifies fp32 rounding, 1.8e-6 measured)
            assert np.isfinite(host(b)).all() and rel_l2(host(b), host(a)) < 1e-5, H
        # (2) a weight that does not fit: every tile takes the fp32 products -> the default kernel's numbers
        wb = field_params(L=16, F=2, lg=12, H=H, mn=16, mx=1024)
        wb.feat_w[1][3, 5] = 600.0
        f32, prs = both(to_spec(ops, wb))
        for a, b in zip(f32, prs):
            assert np.isfinite(host(b)).all() and rel_l2(host(b), host(a)) < 1e-6, H
        # (3) an untrained field: every input of the first layer is ~1e-4
        small = field_params(L=16, F=2, lg=12, H=H, mn=16, mx=1024, scale=1e-4)
        f32, prs = both(to_spec(ops, small))
        for a, b in zip(f32, prs):
            assert rel_l2(host(b), host(a)) < 2e-6, H


def test_fp16_pair_products_in_the_training_forward(switches):
    """Round 6: the per-sample kernel (nrhip_field_fwd / nrhip_field_fwd_train: the training forward that stores its
    activations) can form its matrix products as fp16 pairs too (NRHIP_MLP_PAIRS_TRAIN=1; opt-in: it measured no faster,
    the kernel is bound by its stores and gathers).  The tile runs in units of 2^6; every store -- the three outputs
    and the four saved activation tensors the backward reads -- undoes that exactly.  Held to the fp32-MFMA kernel
    (the default) on everything it writes, fp32 and fp16-storage tables, both widths, and through the exits of the fast
    path (activations / weights that do not fit an fp16 pair)."""
    from neurad_studio_amd import ops

    R, S = 200, 32
    o, d, area, s, e, _ = _sample_rays(R, S, seed=13)
    rays = (dev(o), dev(d), dev(area), dev(s), dev(e))

    def both(fs):
        switches.unset("NRHIP_MLP_PAIRS_TRAIN")
        a = ops.field_fwd_train(fs, *rays)
        a2 = ops.field_fwd(fs, *rays)
        switches.set("NRHIP_MLP_PAIRS_TRAIN", "1")
        b = ops.field_fwd_train(fs, *rays)
        b2 = ops.field_fwd(fs, *rays)
        switches.unset("NRHIP_MLP_PAIRS_TRAIN")
        return (*a[0], *a[1], *a2), (*b[0], *b[1], *b2)

    names = ("feature", "sdf", "head", "enc", "geo_hidden", "feat_in", "feat_hidden", "feature (field_fwd)", "sdf", "head")
    for L, F, mn, mx, H in ((8, 4, 32, 8192, 32), (16, 2, 16, 1024, 64), (8, 4, 32, 8192, 64), (16, 2, 16, 1024, 32)):
        p = field_params(L=L, F=F, lg=12, H=H, mn=mn, mx=mx)
        for half in (False, True):
            f32, prs = both(to_spec(ops, p, half=half))
            for n, a, b in zip(names, f32, prs):
                assert a.shape == b.shape and rel_l2(host(b), host(a)) < 1e-6, (L, F, H, half, n, rel_l2(host(b), host(a)))
            assert torch.equal(prs[3], f32[3])  # the encoding rows: computed before any product, scaled by 2^6 and back
            assert not torch.equal(prs[0], f32[0])  # (a different kernel did run)
    for H in (32, 64):
        big = field_params(L=8, F=4, lg=12, H=H, mn=32, mx=8192)
        big.grid.table[::2] *= 3.0e3  # activations beyond the pair's range in some tiles
        f32, prs = both(to_spec(ops, big))
        for n, a, b in zip(names, f32, prs):
            assert np.isfinite(host(b)).all() and rel_l2(host(b), host(a)) < 1e-5, (H, n)
        wb = field_params(L=8, F=4, lg=12, H=H, mn=32, mx=8192)
        wb.feat_w[1][3, 5] = 600.0  # a weight that does not fit: every tile takes the fp32 products
        f32, prs = both(to_spec(ops, wb))
        for n, a, b in zip(names, f32, prs):
            assert rel_l2(host(b), host(a)) < 1e-6, (H, n)
        small = field_params(L=8, F=4, lg=12, H=H, mn=32, mx=8192, scale=1e-4)  # an untrained field
        f32, prs = both(to_spec(ops, small))
        for n, a, b in zip(names, f32, prs):
            assert rel_l2(host(b), host(a)) < 2e-6, (H, n)
