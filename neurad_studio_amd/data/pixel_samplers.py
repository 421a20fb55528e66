"""Patch sampler on the device: the step before ray generation.

``ScaledPatchSampler`` mirrors ``nerfstudio/data/pixel_samplers.py:605-765`` (config fields, ``sample``,
``collate_image_dataset_batch(_list)``, ``update_sampling_weights``) for image batches that live in HBM.  The random
draws stay torch's (``torch.rand((P, 3))`` as in PixelSampler.sample_method, pixel_samplers.py:100-103, so a seeded run
picks the same patches as the reference on the same device); everything downstream of the draws -- centre, ray indices,
pixel-centre coordinates for nrhip_camera_rays, the ground-truth patch gather -- is one launch of ``nrhip_patch_sample``
(csrc/raygen.hip) instead of the reference's meshgrid / stack / strided slice / advanced indexing chain.

``collated["coords"]`` is an extra key: RayGenerator.forward's ``image_coords[y, x]`` (ray_generators.py:41-55), ready for
``cameras.raygen.camera_rays`` without another gather."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Optional

import torch
from torch import Tensor

from .. import _lib
from ..ops import _ptr, _stream


def patch_sample(images: Optional[Tensor], patch_size: int, patch_scale: int, *, uniforms: Optional[Tensor] = None,
                 centers: Optional[Tensor] = None, image_idx: Optional[Tensor] = None, image_shape=None,
                 want_coords: bool = True):
    """images [N,H,W,C] fp32 or uint8 on the device (None with image_shape=(N,H,W): indices only).  Exactly one of
    uniforms [P,3] fp32 in [0,1) / centers [P,3] int64.  Returns (ray_indices [P*patch_size^2, 3] int64, coords
    [P*patch_size^2, 2] fp32 or None, patches [P,K,K,C] or None), K = patch_size * patch_scale."""
    if (uniforms is None) == (centers is None):
        raise ValueError("patch_sample: exactly one of uniforms / centers")
    src = uniforms if uniforms is not None else centers
    if images is not None:
        if images.dim() != 4 or not images.is_contiguous() or images.dtype not in (torch.float32, torch.uint8):
            raise ValueError("patch_sample: images must be a contiguous [N,H,W,C] fp32 or uint8 tensor")
        if images.device != src.device:
            raise ValueError("patch_sample: images and draws must live on the same device")
        n, h, w, c = images.shape
    else:
        (n, h, w), c = image_shape, 1
    if uniforms is not None and (uniforms.dtype != torch.float32 or uniforms.dim() != 2 or uniforms.shape[1] != 3):
        raise ValueError("patch_sample: uniforms must be [P,3] fp32")
    if centers is not None and (centers.dtype != torch.int64 or centers.dim() != 2 or centers.shape[1] != 3):
        raise ValueError("patch_sample: centers must be [P,3] int64")
    src = src.contiguous()
    P, K, dev = src.shape[0], patch_size * patch_scale, src.device
    if image_idx is not None:
        image_idx = image_idx.to(device=dev, dtype=torch.int64).contiguous()
        if image_idx.shape[0] != n:
            raise ValueError("patch_sample: image_idx must have one entry per image")
    rays = torch.empty((P * patch_size * patch_size, 3), device=dev, dtype=torch.int64)
    coords = torch.empty((P * patch_size * patch_size, 2), device=dev) if want_coords else None
    patches = None if images is None else torch.empty((P, K, K, c), device=dev, dtype=images.dtype)
    _lib.call("nrhip_patch_sample", _ptr(src if uniforms is not None else None), _ptr(src if centers is not None else None), P,
              n, h, w, c, patch_size, patch_scale, _ptr(image_idx), _ptr(images),
              0 if images is None or images.dtype == torch.float32 else 1, _ptr(rays), _ptr(coords), _ptr(patches), _stream())
    return rays, coords, patches


@dataclass
class ScaledPatchSamplerConfig:
    """pixel_samplers.py:36-52,605-616 (the fields the patch sampler reads)"""

    num_rays_per_batch: int = 4096
    keep_full_image: bool = False
    patch_scale: int = 1
    """The upsampling ratio between sampled rays and pixel ground truths."""
    patch_size: int = 1
    """The size of sampled patches."""

    def setup(self, **kwargs) -> "ScaledPatchSampler":
        return ScaledPatchSampler(self, **kwargs)


class ScaledPatchSampler:
    def __init__(self, config: ScaledPatchSamplerConfig, num_rays_per_batch: Optional[int] = None,
                 keep_full_image: Optional[bool] = None, **kwargs) -> None:
        self.config = config
        if num_rays_per_batch is not None:
            self.config.num_rays_per_batch = num_rays_per_batch
        if keep_full_image is not None:
            self.config.keep_full_image = keep_full_image
        self.set_num_rays_per_batch(self.config.num_rays_per_batch)
        self.patch_scale, self.patch_size = config.patch_scale, config.patch_size
        self.sampling_weights: Optional[Tensor] = None
        self.sampling_scale = 1
        self.sampling_shape = None

    def set_num_rays_per_batch(self, num_rays_per_batch: int) -> None:
        self.num_rays_per_batch = num_rays_per_batch

    # ---- pixel_samplers.py:752-763 (a handful of elementwise ops once per evaluation round: torch) ----------------
    def update_sampling_weights(self, scores: Tensor, ratio_uniform: float = 0.5, sampling_scale: int = 1) -> None:
        scores = torch.nn.functional.avg_pool2d(scores, sampling_scale, sampling_scale, ceil_mode=True)
        self.sampling_scale, self.sampling_shape = sampling_scale, scores.shape
        scores = scores.flatten()
        self.sampling_weights = ratio_uniform / scores.numel() + (1 - ratio_uniform) * (scores / scores.sum())

    def _weighted_centers(self, n_patches: int, height: int, width: int, rgb_size: int, device) -> Tensor:
        """the sampling-weights branch of sample_method (pixel_samplers.py:728-749): multinomial draw at the pooled
        resolution, jitter back to pixels, clip into the crop range"""
        flat = torch.multinomial(self.sampling_weights, n_patches, replacement=True)
        sh, sw = self.sampling_shape[-2:]
        img, hh, ww = flat // (sh * sw), (flat % (sh * sw)) // sw, (flat % (sh * sw)) % sw
        if self.sampling_scale > 1:
            jitter = torch.randint(0, self.sampling_scale, (n_patches, 2), device=device)
            hh, ww = hh * self.sampling_scale + jitter[:, 0], ww * self.sampling_scale + jitter[:, 1]
        hh = hh.clip(min=rgb_size // 2, max=height - rgb_size // 2 - 1)
        ww = ww.clip(min=rgb_size // 2, max=width - rgb_size // 2 - 1)
        return torch.stack((img, hh, ww), dim=-1)

    @staticmethod
    def _check_extra_keys(batch: Dict) -> None:
        if set(batch.keys()) - {"image", "image_idx"}:
            raise NotImplementedError("Patch sampler not implemented for extra_keys")

    def collate_image_dataset_batch(self, batch: Dict, num_rays_per_batch: int, keep_full_image: bool = False) -> Dict:
        """pixel_samplers.py:634-664 for a stacked [N,H,W,C] image batch"""
        self._check_extra_keys(batch)
        image = batch["image"]
        if not image.is_cuda:
            raise RuntimeError("ScaledPatchSampler (HIP): the image batch must live on the GPU; there is no CPU path")
        n, h, w, _ = image.shape
        n_patches = num_rays_per_batch // (self.patch_size ** 2)
        rgb_size = self.patch_size * self.patch_scale
        if self.sampling_weights is None:
            draws = dict(uniforms=torch.rand((n_patches, 3), device=image.device))
        else:
            draws = dict(centers=self._weighted_centers(n_patches, h, w, rgb_size, image.device))
        rays, coords, patches = patch_sample(image.contiguous(), self.patch_size, self.patch_scale,
                                             image_idx=batch["image_idx"], **draws)
        out = {"indices": rays, "image": patches, "coords": coords}
        if keep_full_image:
            out["full_image"] = batch["image"]
        return out

    def collate_image_dataset_batch_list(self, batch: Dict, num_rays_per_batch: int, keep_full_image: bool = False) -> Dict:
        """pixel_samplers.py:666-694: images of different sizes -- one draw of the image per patch, then one launch per
        image that received patches"""
        self._check_extra_keys(batch)
        images = batch["image"]
        device = images[0].device
        if not images[0].is_cuda:
            raise RuntimeError("ScaledPatchSampler (HIP): the image batch must live on the GPU; there is no CPU path")
        if self.sampling_weights is not None:
            raise AssertionError("sampling_weights not supported for ScaledPatchSampler in list mode")
        n_patches = num_rays_per_batch // (self.patch_size ** 2)
        img_indices, img_counts = torch.unique(torch.randint(0, len(images), (n_patches,), device=device), return_counts=True)
        rays, coords, patches = [], [], []
        for img_idx, count in zip(img_indices.tolist(), img_counts.tolist()):
            u = torch.rand((count, 3), device=device)
            r, c, p = patch_sample(images[img_idx].unsqueeze(0).contiguous(), self.patch_size, self.patch_scale, uniforms=u,
                                   image_idx=batch["image_idx"][img_idx:img_idx + 1])
            rays.append(r), coords.append(c), patches.append(p)
        out = {"indices": torch.cat(rays), "image": torch.cat(patches), "coords": torch.cat(coords)}
        if keep_full_image:
            out["full_image"] = batch["image"]
        return out

    def sample(self, image_batch: Dict) -> Dict:
        """pixel_samplers.py:368-385 (PixelSampler.sample)"""
        if isinstance(image_batch["image"], list):
            return self.collate_image_dataset_batch_list(dict(image_batch.items()), self.num_rays_per_batch,
                                                         keep_full_image=self.config.keep_full_image)
        if isinstance(image_batch["image"], Tensor):
            return self.collate_image_dataset_batch(image_batch, self.num_rays_per_batch,
                                                    keep_full_image=self.config.keep_full_image)
        raise ValueError("image_batch['image'] must be a list or torch.Tensor")


def lidar_point_sample(lidar: Tensor, points_per_lidar: Tensor, num_rays: int, *, shuffle: Tensor, draws: Tensor,
                       lidar_idx: Optional[Tensor] = None):
    """lidar [sum(points_per_lidar), D] fp32 packed point clouds on the device; shuffle [n] int64 (a permutation), draws
    [n, ceil(num_rays / n)] fp64 in [0,1).  Returns (indices [num_rays,2] int64 = (lidar, point), points [num_rays, D])."""
    if lidar.dim() != 2 or lidar.dtype != torch.float32 or not lidar.is_contiguous():
        raise ValueError("lidar_point_sample: lidar must be a contiguous [points, D] fp32 tensor")
    dev, n = lidar.device, int(points_per_lidar.shape[0])
    rpl = -(-num_rays // n)
    if shuffle.dtype != torch.int64 or shuffle.shape != (n,):
        raise ValueError("lidar_point_sample: shuffle must be [n_lidars] int64")
    if draws.dtype != torch.float64 or draws.shape != (n, rpl):
        raise ValueError("lidar_point_sample: draws must be [n_lidars, ceil(num_rays / n_lidars)] fp64")
    npl = points_per_lidar.to(device=dev, dtype=torch.int64).contiguous()
    if lidar_idx is not None:
        lidar_idx = lidar_idx.to(device=dev, dtype=torch.int64).contiguous()
    indices = torch.empty((num_rays, 2), device=dev, dtype=torch.int64)
    points = torch.empty((num_rays, lidar.shape[1]), device=dev)
    _lib.call("nrhip_lidar_point_sample", _ptr(shuffle.contiguous()), _ptr(draws.contiguous()), _ptr(npl), _ptr(lidar_idx),
              _ptr(lidar), n, rpl, lidar.shape[1], num_rays, _ptr(indices), _ptr(points), _stream())
    return indices, points


@dataclass
class LidarPointSamplerConfig:
    """pixel_samplers.py:36-52,474-479"""

    num_rays_per_batch: int = 4096
    keep_full_image: bool = False

    def setup(self, **kwargs) -> "LidarPointSampler":
        return LidarPointSampler(self, **kwargs)


class LidarPointSampler:
    """pixel_samplers.py:482-601 for the packed batch of ``lidar_packed_collate`` (image_lidar_datamanager.py:60-74):
    ``{"lidar": [sum P_i, D], "points_per_lidar": [n], "lidar_idx": [n]}`` resident in HBM.  torch draws the permutation
    and the fp64 uniforms (same calls, same order as the reference); one kernel does the rest."""

    def __init__(self, config: LidarPointSamplerConfig, num_rays_per_batch: Optional[int] = None, **kwargs) -> None:
        self.config = config
        if num_rays_per_batch is not None:
            self.config.num_rays_per_batch = num_rays_per_batch
        self.set_num_rays_per_batch(self.config.num_rays_per_batch)

    def set_num_rays_per_batch(self, num_rays_per_batch: int) -> None:
        self.num_rays_per_batch = num_rays_per_batch

    def collate_image_dataset_batch(self, batch: Dict, num_rays_per_batch: int, keep_full_image: bool = False) -> Dict:
        if keep_full_image:
            raise NotImplementedError("keep_full_image not implemented for lidar")
        extra = set(batch.keys()) - {"lidar", "lidar_idx", "points_per_lidar"}
        if any(batch[k] is not None for k in extra):
            raise NotImplementedError(f"LidarPointSampler (HIP): per-point extras {sorted(extra)} are not gathered")
        lidar = batch["lidar"]
        if not lidar.is_cuda:
            raise RuntimeError("LidarPointSampler (HIP): the point batch must live on the GPU; there is no CPU path")
        n = len(batch["points_per_lidar"])
        shuffle = torch.randperm(n, device=lidar.device)
        draws = torch.rand((n, -(-num_rays_per_batch // n)), device=lidar.device, dtype=torch.float64)
        indices, points = lidar_point_sample(lidar, torch.as_tensor(batch["points_per_lidar"]), num_rays_per_batch,
                                             shuffle=shuffle, draws=draws, lidar_idx=batch["lidar_idx"])
        return {"lidar": points, "indices": indices}

    def sample(self, image_batch: Dict) -> Dict:
        if isinstance(image_batch["lidar"], Tensor):
            return self.collate_image_dataset_batch(image_batch, self.num_rays_per_batch,
                                                    keep_full_image=self.config.keep_full_image)
        raise NotImplementedError("LidarPointSampler (HIP): list-of-scans batches (lidar_variable_res_collate) keep the "
                                  "reference sampler; pack them with lidar_packed_collate")
