"""The fused render kernel's variants (nrhip_render_fwd_ex: tile-serial / software-pipelined gathers / pipelined with the
last feature layer applied once per ray) against the CPU oracle and against each other, plus the early-ray-termination
option: exact when off, bounded by `early_stop_eps` when on."""
import numpy as np
import pytest
import torch

import neurad_oracle as O
from conftest import rel_l2
from test_gpu_parity import RENDER_CFGS, TOL, _sample_rays, dev, field_params, host, to_spec

pytestmark = pytest.mark.gpu

VARIANTS = {"serial": 1, "pipelined": 2, "pipelined_deferred": 3}


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from neurad_studio_amd import ops as _ops

    return _ops


@pytest.mark.parametrize("variant", list(VARIANTS))
@pytest.mark.parametrize("cfg", RENDER_CFGS)
def test_render_variant_vs_oracle(ops, cfg, variant):
    L, F, lg, mn, mx, H, use_sdf, R, S = cfg
    p = field_params(use_sdf=use_sdf, L=L, F=F, lg=lg, H=H, mn=mn, mx=mx, scale=2.0 if use_sdf else 0.5)
    if use_sdf:
        p.beta = 3.0
    fs = to_spec(ops, p)
    o, d, area, s, e, eu = _sample_rays(R, S, seed=R + S)
    ref = O.render_rays(p, o, d, area, s, e)
    edges = dev(eu)
    feats, depth, acc, w = ops.render_fwd(fs, dev(o), dev(d), dev(area), edges[:, :-1], edges[:, 1:],
                                          return_weights=True, variant=VARIANTS[variant])
    assert rel_l2(host(w), ref["weights"]) < TOL
    assert rel_l2(host(feats), ref["features"]) < TOL
    assert rel_l2(host(depth), ref["depth"]) < TOL or np.abs(host(depth) - ref["depth"]).max() < 1e-5
    assert rel_l2(host(acc), ref["accumulation"]) < TOL


def test_render_variants_agree_more_rays_than_waves(ops):
    """R far above the persistent grid's wave count: every wave walks several rays, the flattened (ray, tile) pipeline
    crosses ray boundaries, ragged S.  All variants must agree to fp32 reassociation."""
    p = field_params(use_sdf=True, L=16, F=2, lg=14, H=64, mn=16, mx=1024, scale=1.0)
    p.beta = 2.0
    fs = to_spec(ops, p)
    R, S = 9000, 37
    o, d, area, s, e, eu = _sample_rays(R, S, seed=3)
    edges = dev(eu)
    outs = {k: ops.render_fwd(fs, dev(o), dev(d), dev(area), edges[:, :-1], edges[:, 1:], return_weights=True, variant=v)
            for k, v in VARIANTS.items()}
    base = outs["serial"]
    for k in ("pipelined", "pipelined_deferred"):
        for a, b in zip(outs[k], base):
            assert rel_l2(host(a), host(b)) < 5e-6, k
    assert torch.equal(outs["pipelined"][3], base[3])  # weights: same arithmetic in the same order
    sl = slice(8990, 9000)  # the tail rays (last pipeline stages) against the oracle
    ref = O.render_rays(p, o[sl], d[sl], area[sl], s[sl], e[sl])
    assert rel_l2(host(outs["pipelined_deferred"][0][sl]), ref["features"]) < TOL


@pytest.mark.parametrize("use_sdf", [True, False])
@pytest.mark.parametrize("variant", [2, 3])
def test_early_ray_termination_bounded(ops, use_sdf, variant):
    """early_stop_eps: rays stop once their transmittance is below eps.  What is dropped weighs < eps in total, so
    accumulation changes by < eps, features by < eps * max|feature|; weights of skipped samples come back as zeros and
    everything before the cut is bit-identical.  eps = 0 is the exact path."""
    p = field_params(use_sdf=use_sdf, L=8, F=4, lg=11, H=32, scale=2.0 if use_sdf else 0.5)
    if use_sdf:
        p.beta = 6.0  # alphas around 0.5: transmittance falls below 1e-3 after a dozen samples
    fs = to_spec(ops, p)
    R, S = 300, 96
    o, d, area, s, e, eu = _sample_rays(R, S, seed=21)
    if not use_sdf:  # make the medium dense enough to saturate
        p.geo_b[1][0] += 4.0
        fs = to_spec(ops, p)
    args = (fs, dev(o), dev(d), dev(area), dev(s), dev(e))
    f0, d0, a0, w0 = ops.render_fwd(*args, return_weights=True, variant=variant)
    eps = 1e-3
    f1, d1, a1, w1 = ops.render_fwd(*args, return_weights=True, variant=variant, early_stop_eps=eps)
    n_skipped = int((w1 == 0).sum() - (w0 == 0).sum())
    assert n_skipped > R * S // 4, "the test scene must actually terminate rays early"
    kept = w1 != 0
    assert torch.equal(w1[kept], w0[kept])
    assert float((w0 * (~kept)).sum(-1).max()) <= eps * 1.01  # per ray: what was skipped weighs < eps
    assert float((a1 - a0).abs().max()) <= eps * 1.01
    fmax = float(f0.abs().max()) + 1.0
    assert float((f1 - f0).abs().max()) <= 3 * eps * fmax
    # the sky residual (1 - acc on the last sample) is dropped with the tail: it is < eps on a terminated ray
    assert float((d1 - d0).abs().max()) <= eps * float(e.max())
    # a threshold nothing reaches: same arithmetic, same result
    f2, d2, a2, w2 = ops.render_fwd(*args, return_weights=True, variant=variant, early_stop_eps=1e-30)
    assert torch.equal(f2, f0) and torch.equal(w2, w0) and torch.equal(a2, a0)


def test_early_stop_rejected_by_serial_kernel(ops):
    from neurad_studio_amd._lib import NeuradHipError

    p = field_params()
    fs = to_spec(ops, p)
    o, d, area, s, e, _ = _sample_rays(8, 32, seed=1)
    with pytest.raises(NeuradHipError):
        ops.render_fwd(fs, dev(o), dev(d), dev(area), dev(s), dev(e), variant=1, early_stop_eps=1e-3)
    with pytest.raises(NeuradHipError):
        ops.render_fwd(fs, dev(o), dev(d), dev(area), dev(s), dev(e), variant=7)
