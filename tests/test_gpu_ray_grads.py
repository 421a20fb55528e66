"""dL/d(origins, directions): the gradient a camera optimizer that moves the rays receives
(cameras/camera_optimizers.py:173-182; `neurad-scaleopt`, configs/method_configs.py:438-447), against the reference's own
autograd (tests/golden/ray_grads.npz, oracle/make_golden_raygrads.py) and the numpy oracle's analytic restatement."""
import numpy as np
import pytest
import torch

import neurad_oracle as O
import synth
from conftest import load_golden, rel_l2
from test_gpu_modules import bundle, dev, host, make_field, make_prop

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _field_grid(lg=11, dtype=np.float32):
    return O.GridParams(synth.hash_table(8 * 2**lg, 4, seed=51, scale=0.5).astype(dtype).astype(np.float32), 8, 32, 8192, lg)


def test_encode_bwd_rays_vs_reference_autograd():
    from neurad_studio_amd import ops

    g = load_golden("ray_grads")
    grid = _field_grid()
    spec = ops.GridSpec(8, 4, 11, 32, 8192)
    args = [dev(g[k]) for k in ("o", "d", "area", "starts", "ends")]
    go, gd = ops.encode_bwd_rays(spec, dev(grid.table), 100.0, *args, dev(g["g_enc"]))
    assert rel_l2(host(go), g["enc_go"]) < TOL and rel_l2(host(gd), g["enc_gd"]) < TOL
    # run to run: no atomics -> bit-identical
    go2, gd2 = ops.encode_bwd_rays(spec, dev(grid.table), 100.0, *args, dev(g["g_enc"]))
    assert torch.equal(go, go2) and torch.equal(gd, gd2)
    # rows that are exactly zero (samples behind an opaque surface / overridden by an actor) contribute nothing
    ge = g["g_enc"].copy()
    ge[::3] = 0
    want = O.encode_static_ray_grads(grid, 100.0, g["o"], g["d"], g["area"], g["starts"], g["ends"], ge)
    go, gd = ops.encode_bwd_rays(spec, dev(grid.table), 100.0, *args, dev(ge))
    assert rel_l2(host(go), want[0]) < TOL and rel_l2(host(gd), want[1]) < TOL


@pytest.mark.parametrize("S", [7, 16, 33, 128])
def test_encode_bwd_rays_group_sizes_and_fp16_table_vs_oracle(S):
    """every lanes-per-ray variant (16 / 32 / 64 lanes, ragged sample counts), fp32 and fp16-storage tables, strided edges"""
    from neurad_studio_amd import ops

    R = 37
    o, d, area, _ = synth.rays(R, 5)
    _, eu, _ = O.power_sampler(np.zeros(R), np.full(R, 3000.0, np.float32), S)
    ge = synth.normal((R * S, 32), seed=9)
    spec = ops.GridSpec(8, 4, 11, 32, 8192)
    edges = dev(eu)
    for dtype, tdt in ((np.float32, torch.float32), (np.float16, torch.float16)):
        grid = _field_grid(dtype=dtype)
        want = O.encode_static_ray_grads(grid, 100.0, o, d, area, eu[:, :-1], eu[:, 1:], ge)
        go, gd = ops.encode_bwd_rays(spec, dev(grid.table).to(tdt), 100.0, dev(o), dev(d), dev(area), edges[:, :-1],
                                     edges[:, 1:], dev(ge))
        assert rel_l2(host(go), want[0]) < TOL and rel_l2(host(gd), want[1]) < TOL, (S, dtype)


def test_proposal_density_ray_gradients_vs_reference_autograd():
    from neurad_studio_amd import autograd as ag

    g = load_golden("ray_grads")
    p = make_prop(91)
    hg = p.hashgrid
    o, d = dev(g["o"]).requires_grad_(True), dev(g["d"]).requires_grad_(True)
    dens = ag.ProposalDensityFn.apply(hg.static_grid.hash_table, p.density_decoder.weight, hg.static_grid.spec, hg.static_scale,
                                      o, d, dev(g["area"]), dev(g["starts"]), dev(g["ends"]))
    assert rel_l2(host(dens), g["prop_dens"]) < TOL
    (dens * dev(g["prop_g_dens"])).sum().backward()
    assert rel_l2(host(o.grad), g["prop_go"]) < TOL and rel_l2(host(d.grad), g["prop_gd"]) < TOL
    # the fused sampler round (weights + depth from the edges): ray gradients against torch autograd through the
    # operator-level nodes on the same inputs
    edges = torch.cat([dev(g["starts"]), dev(g["ends"])[:, -1:]], -1).contiguous()
    gw = dev(synth.normal(g["starts"].shape, seed=3))

    def run(fused):
        o_, d_ = dev(g["o"]).requires_grad_(True), dev(g["d"]).requires_grad_(True)
        if fused:
            w, dep = ag.ProposalRoundFn.apply(hg.static_grid.hash_table, p.density_decoder.weight, hg.static_grid.spec,
                                              hg.static_scale, o_, d_, dev(g["area"]), edges)
        else:
            dn = ag.ProposalDensityFn.apply(hg.static_grid.hash_table, p.density_decoder.weight, hg.static_grid.spec,
                                            hg.static_scale, o_, d_, dev(g["area"]), edges[:, :-1], edges[:, 1:])
            w, dep = ag.PropWeightsFn.apply(edges, dn)
        ((w * gw).sum() + dep.sum()).backward()
        return host(o_.grad), host(d_.grad)

    (a0, a1), (b0, b1) = run(True), run(False)
    assert rel_l2(a0, b0) < 1e-5 and rel_l2(a1, b1) < 1e-5


@pytest.mark.parametrize("fused_training", [True, False], ids=["fused-train", "operator-train"])
def test_field_ray_gradients_vs_reference_autograd(fused_training):
    """NeuRADField.forward with rays that require grad: both training paths (FieldTrainFn; EncodeFn + operator MLPs) hand
    origins and directions the reference's gradient, and leave the parameter gradients what they were"""
    from neurad_studio_amd.field_components.field_heads import FieldHeadNames
    from neurad_studio_amd.model_components.ray_samplers import PowerSampler

    g = load_golden("ray_grads")
    fld = make_field(True).eval()
    fld.fused_training = fused_training
    rb = bundle(g["o"], g["d"], g["area"])
    rb.origins.requires_grad_(True), rb.directions.requires_grad_(True)
    rs = PowerSampler(num_samples=g["starts"].shape[1], lambda_=-1.0, scaling=0.1).eval()(rb)
    out = fld(rs)
    assert rel_l2(host(out[FieldHeadNames.FEATURE]), g["field_feature"]) < TOL
    ((out[FieldHeadNames.FEATURE] * dev(g["field_g_feature"])).sum()
     + (out[FieldHeadNames.ALPHA][..., 0] * dev(g["field_g_alpha"])).sum()).backward()
    # Through the MLPs the bound is the reference's own: its fp32 model against itself in fp64 differs by 2.4e-2 / 3.5e-2 on
    # these gradients, on EVERY ray (golden field_floor: d lerp / dx is bilinear in the other two offsets, which fp32
    # positions resolve to 5e-4 of a finest-level cell; the per-ray sums cancel heavily, the direction gradient weighs every
    # sample with its distance, up to 2e4 m).  The HIP path follows the fp32 arithmetic op for op and sits an order of
    # magnitude below that floor; what remains are summation order and single ReLU-kink flips (fp32 MFMA vs torch's GEMM).
    # The kernel itself is held to 1e-4 against the reference in test_encode_bwd_rays_vs_reference_autograd.
    for got, key, floor in ((rb.origins.grad, "field_go", g["field_floor"][0]), (rb.directions.grad, "field_gd", g["field_floor"][1])):
        a, c = host(got).astype(np.float64), g[key].astype(np.float64)
        per_ray = np.linalg.norm(a - c, axis=-1) / np.linalg.norm(c, axis=-1)
        assert rel_l2(a, c) < 0.1 * floor, (key, rel_l2(a, c), floor, np.sort(per_ray)[-5:])
        assert np.median(per_ray) < 1e-3, (key, np.sort(per_ray))  # (measured: median 1.1e-4, worst ray 1.2e-3)
    tg = host(fld.hashgrid.static_grid.hash_table.grad).copy()
    # same step with fixed rays: the same parameter gradients, and no ray-gradient kernel is launched
    fld.zero_grad()
    rb2 = bundle(g["o"], g["d"], g["area"])
    rs2 = PowerSampler(num_samples=g["starts"].shape[1], lambda_=-1.0, scaling=0.1).eval()(rb2)
    out2 = fld(rs2)
    ((out2[FieldHeadNames.FEATURE] * dev(g["field_g_feature"])).sum()
     + (out2[FieldHeadNames.ALPHA][..., 0] * dev(g["field_g_alpha"])).sum()).backward()
    assert rel_l2(host(fld.hashgrid.static_grid.hash_table.grad), tg) < 1e-6  # (a batch this small takes the atomic scatter)
    assert rb2.origins.grad is None
