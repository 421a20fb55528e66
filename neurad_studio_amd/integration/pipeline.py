"""Pipeline and data manager of the ``neurad-hip`` method: what binds the pieces `bench.py` times into `ns-train`.

The reference's trainer (engine/trainer.py:535-579) asks its pipeline for ``get_train_loss_dict(step)``; the reference's
``ADPipeline`` (pipelines/ad_pipeline.py:57-100) answers with batches from CPU worker processes
(data/datamanagers/image_lidar_datamanager.py:96-169: pixel sampler + ray generator per worker, a queue, a host->device copy
per step) and, at world_size > 1, wraps the model in DDP (pipelines/base_pipeline.py:304-307: 25 MB buckets,
``find_unused_parameters=True``).  Two subclasses, same config fields, same call surface:

``ADHipDataManager``   training images and lidar scans are cached ONCE in HBM (a PandaSet clip: 480 images x 1920 x 1080 x 3
                       bytes = 3 GB + 80 scans; 288 GB are there); ``next_train`` draws the batch on the device --
                       data/pixel_samplers.py (patch centres -> ray indices, pixel-centre coordinates, ground-truth patches)
                       and cameras/raygen.py (rolling-shutter camera rays, lidar rays) -- and merges the two halves with the
                       reference's own ``_merge_img_lidar``.  No worker processes, no queue, no per-step host->device copy.
                       Eval keeps the reference's loaders.
``ADHipPipeline``      world_size > 1: the model stays bare and ``parallel.data_parallel.GradientSynchronizer`` exchanges the
                       gradients (reduce-scatter + all-gather on the tables' own storage, one coalesced all-reduce for the
                       rest) at the END of every backward pass, so that ``grad_scaler.step`` already sees reduced gradients;
                       ``get_train_loss_dict`` tells the model the batch's [camera; lidar] layout (no boolean-mask gathers).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, Tuple, Type

import torch

import nerfstudio.pipelines.base_pipeline as _ref_base_pipeline
from nerfstudio.cameras.rays import RayBundle
from nerfstudio.data.datamanagers.ad_datamanager import ADDataManager, ADDataManagerConfig
from nerfstudio.data.datamanagers.image_lidar_datamanager import _cache_images, _cache_points, _merge_img_lidar, lidar_packed_collate
from nerfstudio.pipelines.ad_pipeline import ADPipeline, ADPipelineConfig

from ..cameras import raygen
from ..data import pixel_samplers as hip_samplers
from ..parallel.data_parallel import GradientSynchronizer


@dataclass
class ADHipDataManagerConfig(ADDataManagerConfig):
    _target: Type = field(default_factory=lambda: ADHipDataManager)
    device_batches: bool = True
    """Training batches drawn on the GPU from HBM-resident images and scans (False: the reference's worker processes)."""
    num_processes: int = 0
    """(reference field) no worker processes are started while ``device_batches`` is on"""


class ADHipDataManager(ADDataManager):
    config: ADHipDataManagerConfig

    def _device_batches(self) -> bool:
        return bool(self.config.device_batches) and str(self.device).startswith("cuda")

    def setup_train(self):
        if not self._device_batches():
            return super().setup_train()
        assert self.train_dataset is not None
        dev = self.device
        cfg = self.config
        ps = cfg.pixel_sampler
        self.train_pixel_sampler = hip_samplers.ScaledPatchSamplerConfig(
            num_rays_per_batch=cfg.train_num_rays_per_batch, patch_scale=ps.patch_scale, patch_size=ps.patch_size).setup()
        self.train_point_sampler = hip_samplers.LidarPointSamplerConfig(
            num_rays_per_batch=cfg.train_num_lidar_rays_per_batch).setup()
        # the reference's own caching helpers (image_lidar_datamanager.py:351-376), then ONE copy to the device
        self._train_images = self._train_points = None
        if len(self.train_dataset.cameras):
            cached = _cache_images(self.train_dataset, cfg.max_thread_workers, cfg.collate_fn)
            if not isinstance(cached["image"], torch.Tensor):
                raise NotImplementedError("device_batches: training images of different sizes (variable_res_collate); set "
                                          "device_batches=False")
            self._train_images = {"image": cached["image"].to(dev).contiguous(), "image_idx": cached["image_idx"].to(dev)}
            self._train_cameras = self.train_dataset.cameras.to(dev)
        if len(self.train_lidar_dataset.lidars):
            cached = _cache_points(self.train_lidar_dataset, cfg.max_thread_workers, lidar_packed_collate)
            self._train_points = {"lidar": cached["lidar"].to(dev).float().contiguous(),
                                  "points_per_lidar": torch.as_tensor(cached["points_per_lidar"]).to(dev),
                                  "lidar_idx": torch.as_tensor(cached["lidar_idx"]).to(dev)}
            self._train_lidars = self.train_lidar_dataset.lidars.to(dev)
        self.data_procs, self.func_queues, self.data_queue, self.use_mp = [], [], None, False

    def next_train(self, step: int) -> Tuple[RayBundle, Dict]:
        if not self._device_batches():
            return super().next_train(step)
        self.train_count += 1
        img_batch = img_bundle = lidar_batch = lidar_bundle = None
        if self._train_images is not None:
            img_batch = self.train_pixel_sampler.sample(self._train_images)
            idx = img_batch["indices"]
            img_bundle = raygen.camera_rays(self._train_cameras, idx[:, 0:1], img_batch.pop("coords"), bundle_cls=RayBundle)
        if self._train_points is not None:
            lidar_batch = self.train_point_sampler.sample(self._train_points)
            lidar_bundle = raygen.lidar_rays(self._train_lidars, lidar_batch.pop("indices")[:, 0:1], lidar_batch["lidar"],
                                             bundle_cls=RayBundle)
        return _merge_img_lidar(img_bundle, img_batch, lidar_bundle, lidar_batch, len(self.train_dataset))

    def change_patch_sampler(self, patch_scale: int, patch_size: int):
        if not self._device_batches():
            return super().change_patch_sampler(patch_scale, patch_size)
        self.train_pixel_sampler.patch_scale = self.train_pixel_sampler.config.patch_scale = patch_scale
        self.train_pixel_sampler.patch_size = self.train_pixel_sampler.config.patch_size = patch_size
        if getattr(self, "eval_pixel_sampler", None) is not None:
            self.eval_pixel_sampler.patch_scale, self.eval_pixel_sampler.patch_size = patch_scale, patch_size

    def clear_data_queue(self):
        if self._device_batches():
            self.next_batch = None
            return
        return super().clear_data_queue()


@dataclass
class ADHipPipelineConfig(ADPipelineConfig):
    _target: Type = field(default_factory=lambda: ADHipPipeline)
    datamanager: ADHipDataManagerConfig = field(default_factory=ADHipDataManagerConfig)
    overlap_grad_exchange: bool = True
    """world_size > 1, scenes without actors: the large table gradients' reduce-scatter starts from their gradient hooks,
    under the rest of the backward (GradientSynchronizer(overlap=True))"""
    wire_bf16: bool = False
    """world_size > 1: reduce-scatter leg of the large fp32 table gradients in bf16 (one rounding per rank, fp32 sum)"""


class _BareModel(_ref_base_pipeline.DDP):
    """stands in for DistributedDataParallel while VanillaPipeline.__init__ runs (pipelines/base_pipeline.py:304-307):
    "constructing" it returns the model as it is (``module_wrapper``'s isinstance test keeps working: the name is still a
    class); ADHipPipeline installs the GradientSynchronizer afterwards"""

    def __new__(cls, model, *args, **kwargs):
        return model


class ADHipPipeline(ADPipeline):
    config: ADHipPipelineConfig

    def __init__(self, config: ADHipPipelineConfig, **kwargs):
        ddp = _ref_base_pipeline.DDP
        _ref_base_pipeline.DDP = _BareModel
        try:
            super().__init__(config, **kwargs)
        finally:
            _ref_base_pipeline.DDP = ddp
        self.grad_sync = None
        if self.world_size > 1:
            params = [p for p in self.model.parameters() if p.requires_grad]
            dynamic = self.model.field.hashgrid.has_actors()  # actor grids receive gradients only when a ray hits them
            self.grad_sync = GradientSynchronizer(
                params, average=True, usage="dynamic" if dynamic else "static",
                overlap=bool(config.overlap_grad_exchange) and not dynamic, auto_sync=True,
                wire_dtype=torch.bfloat16 if config.wire_bf16 else None)

    def get_train_loss_dict(self, step: int):
        """pipelines/ad_pipeline.py:78-100 with the batch's layout handed to the model first"""
        ray_bundle, batch = self.datamanager.next_train(step)
        if hasattr(self.model, "set_batch_layout") and "lidar" in batch and "image" in batch:
            n_lidar = batch["lidar"].shape[0]
            self.model.set_batch_layout(len(ray_bundle) - n_lidar, n_lidar)  # _merge_img_lidar: [camera rays; lidar rays]
        model_outputs = self._model(ray_bundle, patch_size=self.config.ray_patch_size)
        metrics_dict = self.model.get_metrics_dict(model_outputs, batch)
        actors = self.model.dynamic_actors
        if actors.config.optimize_trajectories:
            pos_norm = (actors.actor_positions - actors.initial_positions).norm(dim=-1)
            moved = pos_norm > 0
            n_moved = moved.sum().clamp_min(1)
            # (the reference's `x[pos_norm > 0].mean().nan_to_num()`: a masked mean without the nonzero + host read)
            metrics_dict["traj_opt_translation"] = (pos_norm * moved).sum() / n_moved * moved.any()
            rot = (actors.actor_rotations_6d - actors.initial_rotations_6d).norm(dim=-1)
            metrics_dict["traj_opt_rotation"] = (rot * moved).sum() / n_moved * moved.any()
        loss_dict = self.model.get_loss_dict(model_outputs, batch, metrics_dict)
        return model_outputs, loss_dict, metrics_dict
