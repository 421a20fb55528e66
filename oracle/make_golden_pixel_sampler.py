"""Golden vectors for the device patch sampler from the reference's ScaledPatchSampler -- build
container only:  python oracle/make_golden_pixel_sampler.py  -> tests/golden/patch_sampler.npz

Each case seeds torch, records the torch.rand((P, 3)) draws PixelSampler.sample_method will consume
(nerfstudio/data/pixel_samplers.py:100-103), re-seeds and lets the reference sample: its ``indices`` / ``image`` are the
golden outputs, the recorded draws the kernel's input.  The ``centers`` case drives ``_patches_from_centers`` (:696-714)
directly, as the sampling-weights branch does after its multinomial draw."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import ref_import

ref_import.install()
import synth  # noqa: E402
from make_golden import save  # noqa: E402
from nerfstudio.data.pixel_samplers import (LidarPointSampler, LidarPointSamplerConfig, ScaledPatchSampler,  # noqa: E402
                                             ScaledPatchSamplerConfig)
from nerfstudio.model_components.ray_generators import RayGenerator  # noqa: E402,F401  (coords = image_coords[y, x])

CASES = {  # tag: (n_images, H, W, patch_size, patch_scale, n_rays)
    "neurad": (2, 40, 64, 8, 3, 8 * 8 * 6),   # the method's shape at reduced size: 3x upsampled rgb patches
    "odd": (2, 40, 56, 5, 1, 5 * 5 * 7),       # odd K: offsets -(K//2) .. K//2
    "even": (4, 33, 47, 2, 2, 2 * 2 * 9),
    "single": (1, 16, 16, 1, 1, 11),           # PixelSampler-like: one pixel per "patch"
}


def main():
    gold = {}
    for tag, (n, h, w, ps, sc, rays) in CASES.items():
        seed = sum(map(ord, tag))
        image = synth.uniform((n, h, w, 3), 0, 1, seed)
        image_idx = (np.arange(n) * 3 + 2).astype(np.int64)
        sampler = ScaledPatchSampler(ScaledPatchSamplerConfig(patch_scale=sc, patch_size=ps), num_rays_per_batch=rays)
        n_patches = rays // ps ** 2
        torch.manual_seed(seed)
        u = torch.rand((n_patches, 3))
        torch.manual_seed(seed)
        out = sampler.sample({"image": torch.from_numpy(image), "image_idx": torch.from_numpy(image_idx)})
        idx = out["indices"]
        # RayGenerator.forward's coords: image_coords[y, x] (ray_generators.py:41-55; Cameras.get_image_coords = pixel centres)
        coords = torch.stack(torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij"), -1).float() + 0.5
        gold.update({f"{tag}_image": image, f"{tag}_image_idx": image_idx, f"{tag}_uniforms": u.numpy(),
                     f"{tag}_indices": idx.numpy(), f"{tag}_patches": out["image"].numpy(),
                     f"{tag}_coords": coords[idx[:, 1], idx[:, 2]].numpy(),
                     f"{tag}_shape": np.array([n, h, w, ps, sc], dtype=np.int64)})
    # centres handed over (sampling-weights branch): incl. the extreme legal centres
    n, h, w, ps, sc = 2, 30, 44, 3, 2
    K = ps * sc
    image = synth.uniform((n, h, w, 3), 0, 1, 77)
    centers = np.array([[0, K // 2, K // 2], [1, h - K // 2 - 1, w - K // 2 - 1], [1, 15, 20], [0, K // 2, w - K // 2 - 1]],
                       dtype=np.int64)
    sampler = ScaledPatchSampler(ScaledPatchSamplerConfig(patch_scale=sc, patch_size=ps), num_rays_per_batch=ps * ps * 4)
    rays, patches = sampler._patches_from_centers(torch.from_numpy(image), torch.from_numpy(centers), K, "cpu")
    gold.update(centers_image=image, centers_centers=centers, centers_indices=rays.numpy(), centers_patches=patches.numpy(),
                centers_shape=np.array([n, h, w, ps, sc], dtype=np.int64))
    # LidarPointSampler on packed scans (pixel_samplers.py:538-583): torch.randperm then torch.rand(float64), recorded
    for tag, sizes, rays in (("lidar", [37, 5, 120, 64, 1], 203), ("lidar_one", [50], 16)):
        seed = sum(map(ord, tag))
        npl = np.array(sizes, dtype=np.int64)
        cloud = synth.normal((int(npl.sum()), 5), seed)
        lidar_idx = (np.arange(len(sizes)) * 2 + 1).astype(np.int64)
        sampler = LidarPointSampler(LidarPointSamplerConfig(), num_rays_per_batch=rays)
        rpl = -(-rays // len(sizes))
        torch.manual_seed(seed)
        perm = torch.randperm(len(sizes))
        draws = torch.rand((len(sizes), rpl), dtype=torch.float64)
        torch.manual_seed(seed)
        out = sampler.sample({"lidar": torch.from_numpy(cloud), "lidar_idx": torch.from_numpy(lidar_idx),
                              "points_per_lidar": torch.from_numpy(npl)})
        gold.update({f"{tag}_cloud": cloud, f"{tag}_points_per_lidar": npl, f"{tag}_lidar_idx": lidar_idx,
                     f"{tag}_shuffle": perm.numpy(), f"{tag}_draws": draws.numpy(), f"{tag}_indices": out["indices"].numpy(),
                     f"{tag}_points": out["lidar"].numpy()})
    save("patch_sampler", **gold)


if __name__ == "__main__":
    main()
