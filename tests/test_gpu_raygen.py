"""Device ray generation (csrc/raygen.hip) against the reference's own generators
(tests/golden/raygen.npz from Cameras.generate_rays / Lidars.generate_rays, oracle/make_golden_raygen.py)."""
import types

import numpy as np
import pytest
import torch

from conftest import load_golden, rel_l2

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    return t.detach().cpu().numpy()


@pytest.mark.parametrize("mode", ["rows", "cols", "cols_rev"])
def test_camera_rays_vs_reference(mode):
    from neurad_studio_amd.cameras.raygen import camera_rays

    g = load_golden("raygen")
    C = g["c2w"].shape[0]
    md = {"rolling_shutter_time": dev(g["rolling_shutter_time"]), "time_to_center_pixel": dev(g["time_to_center_pixel"]),
          "velocities": dev(g["cam_velocities"]), "sensor_idxs": torch.arange(C, device="cuda")[:, None]}
    if mode != "rows":
        md["rs_direction"] = {"cols": "Horizontal", "cols_rev": "Horizontal_reversed"}[mode]
    cams = types.SimpleNamespace(camera_to_worlds=dev(g["c2w"]), fx=dev(g["fx"]), fy=dev(g["fy"]), cx=dev(g["cx"]),
                                 cy=dev(g["cy"]), width=torch.full((C, 1), 1920, device="cuda"),
                                 height=torch.full((C, 1), 1080, device="cuda"), times=dev(g["cam_times"]), metadata=md,
                                 camera_type=torch.ones((C, 1), dtype=torch.long, device="cuda"), distortion_params=None)
    rb = camera_rays(cams, dev(g["cam_idx"])[:, None], dev(g["coords"]))
    # directions feed floor() in every hash level downstream: held to a few ulps, not to the 1e-4 bar
    assert np.abs(host(rb.directions) - g[f"cam_{mode}_directions"]).max() < 3e-7
    assert np.abs(host(rb.origins) - g[f"cam_{mode}_origins"]).max() < 2e-6
    assert rel_l2(host(rb.pixel_area), g[f"cam_{mode}_pixel_area"]) < 2e-4  # a difference of nearly equal unit vectors
    assert np.abs(host(rb.times) - g[f"cam_{mode}_times"]).max() < 1e-6
    assert rel_l2(host(rb.metadata["directions_norm"]), g[f"cam_{mode}_directions_norm"]) < 1e-6
    assert torch.equal(rb.metadata["sensor_idxs"][:, 0], dev(g["cam_idx"])) and "rolling_shutter_time" not in rb.metadata
    assert float(rb.fars.min()) == 1_000_000.0 and rb.camera_indices.shape == (512, 1)
    cams.distortion_params = torch.ones((C, 6), device="cuda")
    with pytest.raises(NotImplementedError):
        camera_rays(cams, dev(g["cam_idx"])[:, None], dev(g["coords"]))


@pytest.mark.parametrize("ego", [True, False])
def test_lidar_rays_vs_reference(ego):
    from neurad_studio_amd.cameras.raygen import lidar_rays

    g = load_golden("raygen")
    tag = "ego" if ego else "noego"
    lid = types.SimpleNamespace(lidar_to_worlds=dev(g["l2w"]), times=dev(g["lidar_times"]),
                                metadata={"velocities": dev(g["lidar_velocities"])},
                                horizontal_beam_divergence=dev(g["hdiv"]), vertical_beam_divergence=dev(g["vdiv"]),
                                assume_ego_compensated=ego, valid_lidar_distance_threshold=1000.0)
    rb = lidar_rays(lid, dev(g["lidar_idx"])[:, None], dev(g["points"]))
    assert np.abs(host(rb.directions) - g[f"lid_{tag}_directions"]).max() < 3e-7
    assert np.abs(host(rb.origins) - g[f"lid_{tag}_origins"]).max() < 2e-6
    assert rel_l2(host(rb.pixel_area), g[f"lid_{tag}_pixel_area"]) < 1e-6
    assert rel_l2(host(rb.metadata["directions_norm"]), g[f"lid_{tag}_distance"]) < 1e-6
    np.testing.assert_array_equal(host(rb.metadata["did_return"]), g[f"lid_{tag}_did_return"])
    assert not bool(rb.metadata["did_return"].all()) and bool(rb.metadata["is_lidar"].all())
    assert np.abs(host(rb.times) - g[f"lid_{tag}_times"]).max() < 1e-6
    # the generated bundle drives the hot path as is
    from neurad_studio_amd import ops

    order = ops.ray_order(rb.origins, rb.directions, 100.0)
    assert order.shape == (600,)
