"""Device patch sampler (csrc/raygen.hip nrhip_patch_sample) against the reference's
ScaledPatchSampler (tests/golden/patch_sampler.npz, oracle/make_golden_pixel_sampler.py) and the numpy restatement.
Integer work and copies: everything is compared bit for bit."""
import numpy as np
import pytest
import torch

import neurad_oracle as O
from conftest import load_golden

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    return t.detach().cpu().numpy()


@pytest.mark.parametrize("tag", ["neurad", "odd", "even", "single"])
def test_patch_sample_vs_reference(tag):
    from neurad_studio_amd.data.pixel_samplers import patch_sample

    g = load_golden("patch_sampler")
    n, h, w, ps, sc = (int(v) for v in g[f"{tag}_shape"])
    rays, coords, patches = patch_sample(dev(g[f"{tag}_image"]), ps, sc, uniforms=dev(g[f"{tag}_uniforms"]),
                                         image_idx=dev(g[f"{tag}_image_idx"]))
    np.testing.assert_array_equal(host(rays), g[f"{tag}_indices"])
    np.testing.assert_array_equal(host(patches), g[f"{tag}_patches"])
    np.testing.assert_array_equal(host(coords), g[f"{tag}_coords"])


def test_patch_sample_given_centers_vs_reference():
    from neurad_studio_amd.data.pixel_samplers import patch_sample

    g = load_golden("patch_sampler")
    n, h, w, ps, sc = (int(v) for v in g["centers_shape"])
    rays, coords, patches = patch_sample(dev(g["centers_image"]), ps, sc, centers=dev(g["centers_centers"]), want_coords=False)
    assert coords is None
    np.testing.assert_array_equal(host(rays), g["centers_indices"])
    np.testing.assert_array_equal(host(patches), g["centers_patches"])


def test_patch_sample_full_size_uint8_vs_oracle():
    """the method's own shape (models/neurad.py: 32x32 ray patches, 3x upsampled rgb; 40 patches = 40 960 rays,
    pipelines/ad_pipeline.py:49) on uint8 1920x1280 frames, incl. draws at both ends of [0, 1)"""
    from neurad_studio_amd.data.pixel_samplers import patch_sample

    n, h, w, ps, sc, P = 6, 1280, 1920, 32, 3, 40
    gen = torch.Generator(device="cuda").manual_seed(3)
    images = torch.randint(0, 256, (n, h, w, 3), device="cuda", dtype=torch.uint8, generator=gen)
    u = torch.rand((P, 3), device="cuda", generator=gen)
    u[0] = 0.0
    u[1] = float(np.nextafter(np.float32(1.0), np.float32(0.0)))
    image_idx = torch.arange(n, device="cuda") + 100
    rays, coords, patches = patch_sample(images, ps, sc, uniforms=u, image_idx=image_idx)
    c = O.patch_centers_from_uniforms(host(u), n, h, w, ps * sc)
    assert c[:, 0].max() < n and c[:, 1].max() + 48 <= h and c[:, 2].max() + 48 <= w  # the last fp32 below 1 stays inside
    r_ref, c_ref, p_ref = O.patches_from_centers(host(images), c, ps, sc, image_idx=host(image_idx))
    np.testing.assert_array_equal(host(rays), r_ref)
    np.testing.assert_array_equal(host(coords), c_ref)
    np.testing.assert_array_equal(host(patches), p_ref)
    assert rays.shape == (40960, 3) and patches.shape == (P, 96, 96, 3) and patches.dtype == torch.uint8


def test_scaled_patch_sampler_class_seeded_and_feeds_camera_rays():
    """the mirror class draws with torch.rand like the reference: re-seeding reproduces its batch, which equals the
    restatement on those draws; ``coords`` are the pixel centres of ``indices``; list mode and the weighted branch run"""
    from neurad_studio_amd.data.pixel_samplers import ScaledPatchSampler, ScaledPatchSamplerConfig

    n, h, w, ps, sc = 3, 64, 96, 4, 3
    images = torch.rand((n, h, w, 3), device="cuda")
    image_idx = torch.tensor([7, 8, 11], device="cuda")
    s = ScaledPatchSamplerConfig(patch_size=ps, patch_scale=sc).setup(num_rays_per_batch=ps * ps * 10)
    torch.manual_seed(5)
    out = s.sample({"image": images, "image_idx": image_idx})
    torch.manual_seed(5)
    u = torch.rand((10, 3), device="cuda")
    c = O.patch_centers_from_uniforms(host(u), n, h, w, ps * sc)
    r_ref, c_ref, p_ref = O.patches_from_centers(host(images), c, ps, sc, image_idx=host(image_idx))
    np.testing.assert_array_equal(host(out["indices"]), r_ref)
    np.testing.assert_array_equal(host(out["image"]), p_ref)
    np.testing.assert_array_equal(host(out["coords"]), c_ref)
    # list mode: images of different sizes
    lst = [torch.rand((40, 50, 3), device="cuda"), torch.rand((64, 48, 3), device="cuda")]
    out = s.sample({"image": lst, "image_idx": torch.tensor([3, 9], device="cuda")})
    idx, img = host(out["indices"]), host(out["image"])
    assert idx.shape == (160, 3) and img.shape == (10, 12, 12, 3) and set(np.unique(idx[:, 0])) <= {3, 9}
    for p in range(10):  # every patch is the crop its ray indices describe
        src = host(lst[0 if idx[p * 16, 0] == 3 else 1])
        y0, x0 = idx[p * 16, 1] - sc // 2, idx[p * 16, 2] - sc // 2
        np.testing.assert_array_equal(img[p], src[y0:y0 + 12, x0:x0 + 12])
    # sampling-weights branch: all mass on one pooled cell of image 1 -> every patch sits there (after the clip)
    scores = torch.zeros((n, h, w), device="cuda")
    scores[1, 20:24, 40:44] = 1.0
    s.update_sampling_weights(scores, ratio_uniform=0.0, sampling_scale=4)
    out = s.sample({"image": images, "image_idx": image_idx})
    idx = host(out["indices"])
    assert (idx[:, 0] == 8).all() and idx[:, 1].min() >= 20 - 6 and idx[:, 1].max() < 24 + 6
    assert idx[:, 2].min() >= 40 - 6 and idx[:, 2].max() < 44 + 6


def test_patch_sample_errors():
    from neurad_studio_amd._lib import NeuradHipError
    from neurad_studio_amd.data.pixel_samplers import patch_sample

    img = torch.rand((1, 8, 8, 3), device="cuda")
    with pytest.raises(NeuradHipError):  # rgb patch larger than the image
        patch_sample(img, 4, 3, uniforms=torch.rand((2, 3), device="cuda"))
    with pytest.raises(ValueError):
        patch_sample(img, 2, 1)
    r, c, p = patch_sample(img, 2, 1, uniforms=torch.rand((0, 3), device="cuda"))  # empty batch
    assert r.shape == (0, 3) and p.shape == (0, 2, 2, 3)


@pytest.mark.parametrize("tag,rays", [("lidar", 203), ("lidar_one", 16)])
def test_lidar_point_sample_vs_reference(tag, rays):
    from neurad_studio_amd.data.pixel_samplers import lidar_point_sample

    g = load_golden("patch_sampler")
    idx, pts = lidar_point_sample(dev(g[f"{tag}_cloud"]), dev(g[f"{tag}_points_per_lidar"]), rays,
                                  shuffle=dev(g[f"{tag}_shuffle"]), draws=dev(g[f"{tag}_draws"]),
                                  lidar_idx=dev(g[f"{tag}_lidar_idx"]))
    np.testing.assert_array_equal(host(idx), g[f"{tag}_indices"])
    np.testing.assert_array_equal(host(pts), g[f"{tag}_points"])


def test_lidar_point_sampler_class_full_size_feeds_lidar_rays():
    """16 384 lidar rays per batch (BASELINE config[3]) from 300 packed scans; seeded class run ==
    the restatement on the same draws; the gathered points go straight into nrhip_lidar_rays"""
    import types

    from neurad_studio_amd.cameras.raygen import lidar_rays
    from neurad_studio_amd.data.pixel_samplers import LidarPointSamplerConfig

    n, rays = 300, 16384
    gen = torch.Generator(device="cuda").manual_seed(1)
    npl = torch.randint(5000, 15000, (n,), device="cuda", generator=gen)
    cloud = torch.randn((int(npl.sum()), 5), device="cuda", generator=gen) * 20
    cloud[:, 4] = torch.rand((cloud.shape[0],), device="cuda", generator=gen) * 0.1
    lidar_idx = torch.arange(n, device="cuda")
    s = LidarPointSamplerConfig().setup(num_rays_per_batch=rays)
    torch.manual_seed(9)
    out = s.sample({"lidar": cloud, "lidar_idx": lidar_idx, "points_per_lidar": npl.cpu()})
    torch.manual_seed(9)
    perm = torch.randperm(n, device="cuda")
    draws = torch.rand((n, -(-rays // n)), device="cuda", dtype=torch.float64)
    i_ref, p_ref = O.lidar_point_sample(host(cloud), host(npl), rays, host(perm), host(draws), lidar_idx=host(lidar_idx))
    np.testing.assert_array_equal(host(out["indices"]), i_ref)
    np.testing.assert_array_equal(host(out["lidar"]), p_ref)
    l2w = torch.eye(4, device="cuda")[:3][None].repeat(n, 1, 1).contiguous()
    lidars = types.SimpleNamespace(lidar_to_worlds=l2w, times=torch.zeros((n, 1), device="cuda"), metadata=None,
                                   horizontal_beam_divergence=torch.full((n, 1), 3e-3, device="cuda"),
                                   vertical_beam_divergence=torch.full((n, 1), 1.5e-3, device="cuda"),
                                   assume_ego_compensated=True, valid_lidar_distance_threshold=1e3)
    rb = lidar_rays(lidars, out["indices"][:, 0:1], out["lidar"])
    assert rb.origins.shape == (rays, 3) and torch.isfinite(rb.directions).all()
