"""S6: occupancy-grid march (ballot/popcount compaction) vs its CPU restatement + invariants.  Parity with nerfacc is
unpinned (un-vendored, no caller, no golden in the reference) -- see DESIGN.md."""
import numpy as np
import pytest
import torch

import occgrid_oracle as OO
import synth

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("cone", [0.0, 0.01])
def test_occgrid_march_vs_restatement(cone):
    from neurad_studio_amd import ops

    res, R = 32, 200
    rng = np.random.default_rng(0)
    binaries = rng.random((res, res, res)) < 0.3
    aabb = np.array([-10, -10, -2, 10, 10, 6], np.float32)
    o = (synth.normal((R, 3), 1) * np.array([6.0, 6.0, 2.0])).astype(np.float32)
    d = synth.normal((R, 3), 2)
    d[:5, 0] = 0.0  # axis-parallel components
    d = (d / np.linalg.norm(d, axis=-1, keepdims=True)).astype(np.float32)
    t_max = synth.uniform((R,), 5.0, 60.0, 3)
    ri, ts, te, seg = ops.occgrid_march(ops.OccGridSpec(torch.from_numpy(aabb), dev(binaries)), dev(o), dev(d), 0.25,
                                        near_plane=0.1, far_plane=40.0, t_max=dev(t_max), cone_angle=cone)
    rri, rts, rte = OO.occgrid_march(aabb, binaries, o, d, 0.25, 0.1, 40.0, None, t_max, cone)
    ri, ts, te = ri.cpu().numpy(), ts.cpu().numpy(), te.cpu().numpy()
    # midpoint cell tests are discrete: allow a handful of boundary flips, compare the rest exactly
    assert abs(len(ri) - len(rri)) <= 3
    if len(ri) == len(rri):
        np.testing.assert_array_equal(ri, rri)
        assert np.abs(ts - rts).max() < 1e-4 and np.abs(te - rte).max() < 1e-4
    # invariants: packed order, segments, sorted non-overlapping intervals inside [near, min(far, t_max)]
    assert np.all(np.diff(ri) >= 0) and seg[-1].item() == len(ri)
    for r in np.unique(ri)[:50]:
        m = ri == r
        assert np.all(ts[m][1:] >= te[m][:-1] - 1e-5) and np.all(te[m] > ts[m])
        assert ts[m].min() >= 0.1 - 1e-6 and te[m].max() <= min(40.0, t_max[r]) + 1e-5
    # every emitted midpoint sits in an occupied cell
    p = o[ri] + d[ri] * (0.5 * (ts + te))[:, None]
    idx = np.clip(np.floor((p - aabb[:3]) / (aabb[3:] - aabb[:3]) * res).astype(int), 0, res - 1)
    assert binaries[idx[:, 0], idx[:, 1], idx[:, 2]].mean() > 0.995


def test_estimator_sampling_with_alpha_pruning_and_empty_grid():
    from neurad_studio_amd import ops
    from neurad_studio_amd.shims.nerfacc import OccGridEstimator

    est = OccGridEstimator([-5, -5, -5, 5, 5, 5], resolution=16)
    R = 64
    o = dev(np.zeros((R, 3), np.float32))
    d = synth.normal((R, 3), 5)
    d = dev((d / np.linalg.norm(d, axis=-1, keepdims=True)).astype(np.float32))
    ri, ts, te = est.sampling(o, d, render_step_size=0.5, far_plane=100.0)
    assert ri.numel() > 0 and float(te.max()) <= 5 * np.sqrt(3) + 1e-3
    alpha_fn = lambda ts, te, ri: torch.full_like(ts, 0.5)  # noqa: E731
    ri2, ts2, te2 = est.sampling(o, d, alpha_fn=alpha_fn, render_step_size=0.5, far_plane=100.0, early_stop_eps=1e-2)
    # T = 0.5^k >= 1e-2  ->  at most 7 samples per ray survive
    assert ri2.numel() <= 7 * R and ri2.numel() < ri.numel()
    a = torch.full((ri.numel(),), 0.5, device="cuda")
    seg = torch.zeros(R + 1, dtype=torch.int64, device="cuda")
    seg[1:] = torch.cumsum(torch.bincount(ri, minlength=R), 0)
    keep = ops.packed_visibility_from_alpha(a, seg, 1e-2, 0.0).cpu().numpy()
    np.testing.assert_array_equal(keep, OO.packed_visibility_from_alpha(a.cpu().numpy(), seg.cpu().numpy(), 1e-2, 0.0))
    est.binaries[:] = False
    ri3, _, _ = est.sampling(o, d, render_step_size=0.5)
    assert ri3.numel() == 0


def test_volumetric_sampler_module():
    from neurad_studio_amd.cameras.rays import RayBundle
    from neurad_studio_amd.model_components.ray_samplers import VolumetricSampler
    from neurad_studio_amd.shims.nerfacc import OccGridEstimator

    est = OccGridEstimator([-5, -5, -5, 5, 5, 5], resolution=16)
    R = 32
    d = synth.normal((R, 3), 6)
    rb = RayBundle(origins=torch.zeros(R, 3, device="cuda"),
                   directions=dev((d / np.linalg.norm(d, axis=-1, keepdims=True)).astype(np.float32)),
                   pixel_area=torch.full((R, 1), 1e-6, device="cuda"), nears=torch.zeros(R, 1, device="cuda"),
                   fars=torch.full((R, 1), 3.0, device="cuda"))
    s = VolumetricSampler(est).eval()
    rs, ri = s(rb, render_step_size=0.25)
    assert rs.frustums.starts.shape == (ri.shape[0], 1) and float(rs.frustums.ends.max()) <= 3.0 + 1e-5
    assert ri.shape[0] == R * 12  # 3.0 / 0.25 intervals per ray in a fully occupied grid
    with pytest.raises(RuntimeError):
        s.generate_ray_samples()
    est.binaries[:] = False
    rs2, ri2 = s(rb, render_step_size=0.25)
    assert ri2.shape[0] == 1 and float(rs2.frustums.starts[0, 0]) == 1.0  # fake-sample fallback
