"""``neurad-hip``: NeuRAD with its volumetric hot path on libneurad_hip.so, as a nerfstudio method plugin.

    export NERFSTUDIO_METHOD_CONFIGS="neurad-hip=neurad_studio_amd.integration.neurad_hip:neurad_hip"
    ns-train neurad-hip pandaset-data ...

(or list ``neurad_hip`` under the ``nerfstudio.method_configs`` entry-point group -- plugins/registry.py:34-79).

Nothing of neurad-studio is edited or re-typed: ``NeuRADHipModel`` IS the reference's NeuRADModel
(nerfstudio/models/neurad.py:164) -- same config, decoders, losses, metrics, checkpoints -- with
  * ``field`` / ``proposal_fields`` built from this package's NeuRADField / NeuRADProposalField through the configs'
    ``_target`` (configs/base_config.py:47-54); same state_dict names, so neurad checkpoints load;
  * ``sampler``, the renderers, ``lidar_decoder`` and the two sampler losses replaced by their HIP-backed namesakes
    after ``populate_modules``;
  * ``get_nff_outputs`` running the two fused kernels for eval chunks (FusedEvalMixin), the fused training nodes for
    training steps (FusedTrainMixin: sampler rounds, field + head + compositing + appearance as a handful of autograd
    nodes, same output keys; scenes with dynamic actors run on the same nodes with per-sample row overrides) and the
    reference's OWN ``get_nff_outputs`` (models/neurad.py:368-421) otherwise (``fused_training=False``, options the fused
    nodes do not cover).
Per-actor 3-D grids are used (``use_4d_hashgrid=False``): the 4-D grid exists only inside tiny-cuda-nn.
"""
from __future__ import annotations

import contextlib
import dataclasses
from copy import deepcopy
from dataclasses import dataclass
from typing import Literal, Type

import torch

import nerfstudio.models.neurad as _ref_neurad
from nerfstudio.engine.optimizers import AdamOptimizerConfig, AdamWOptimizerConfig
from nerfstudio.field_components.neurad_encoding import ActorSettings, NeuRADHashEncodingConfig
from nerfstudio.fields.neurad_field import NeuRADFieldConfig, NeuRADProposalFieldConfig
from nerfstudio.models.neurad import NeuRADModel, NeuRADModelConfig, SamplingSettings
from nerfstudio.plugins.types import MethodSpecification

from ..field_components.mlp import MLP as HipMLP
from ..fields.neurad_field import NeuRADField as HipNeuRADField
from ..fields.neurad_field import NeuRADProposalField as HipNeuRADProposalField
from ..model_components import losses as hip_losses
from ..model_components import ray_samplers as hip_samplers
from ..model_components import renderers as hip_renderers
from ..models.neurad import FusedEvalMixin, FusedTrainMixin
from ..optim import HashGridAdam
from ..shims import nerfacc as hip_nerfacc


@dataclass
class HashGridAdamConfig(AdamOptimizerConfig):
    """``hashgrids`` group of the method (configs/method_configs.py:423-426: Adam, lr 1e-2, eps 1e-15) on csrc/adam.hip.  The
    reference builds every optimizer as ``config._target(params, **fields)`` (engine/optimizers.py:39-62) and steps it through
    ``grad_scaler.step`` (engine/optimizers.py:160-181): HashGridAdam takes the scale and the found-inf flag from the
    GradScaler on the device (optim.py), so neither the 0.6 GB of table gradients get an unscale pass nor do fp16 gradients
    (``table_dtype="float16"``) trip ``GradScaler.unscale_``."""

    _target: Type = HashGridAdam
    capturable: bool = False
    """step counts on the device from the first step (a training step captured in a HIP graph)"""


def _field_config() -> NeuRADFieldConfig:
    """the reference's default main-field config (fields/neurad_field.py:46-52) -> HIP field, 3-D actor grids"""
    return NeuRADFieldConfig(_target=HipNeuRADField, grid=NeuRADHashEncodingConfig(
        require_actor_grad=True, actor=ActorSettings(flip_prob=0.25, use_4d_hashgrid=False)))


def _proposal_config() -> NeuRADProposalFieldConfig:
    cfg = NeuRADProposalFieldConfig()  # the reference's defaults (fields/neurad_field.py:158-179)
    cfg._target = HipNeuRADProposalField
    cfg.grid.actor.use_4d_hashgrid = False
    return cfg


def _sampling() -> SamplingSettings:
    return SamplingSettings(proposal_field_1=_proposal_config(), proposal_field_2=_proposal_config())


@dataclass
class FusedAdamConfig(AdamOptimizerConfig):
    """the small groups' torch.optim.Adam with ``fused=True``: one multi-tensor launch per group, and -- like HashGridAdam --
    an optimizer that takes the GradScaler's scale / found-inf on the device, where the reference's default (foreach) Adam
    makes ``GradScaler.step`` read found-inf back to the host for every optimizer and iteration (torch/amp/grad_scaler.py
    `_maybe_opt_step`: a ``.item()``)"""

    fused: bool = True


@dataclass
class FusedAdamWConfig(AdamWOptimizerConfig):
    fused: bool = True


@dataclass
class NeuRADHipModelConfig(NeuRADModelConfig):
    _target: Type = dataclasses.field(default_factory=lambda: NeuRADHipModel)
    sampling: SamplingSettings = dataclasses.field(default_factory=_sampling)
    field: NeuRADFieldConfig = dataclasses.field(default_factory=_field_config)  # (shadows dataclasses.field below)
    fused_eval: bool = True
    """Eval chunks through the two fused kernels (False: operator-level path everywhere)."""
    fused_decoder: bool = True
    """decode_features' RGB CNN decoder on csrc/decoder.hip (False: the torch modules)."""
    early_stop_eps: float = 0.0
    """> 0: eval rays stop marching once their transmittance is below it (bounded error); 0 = exact."""
    order_rays: bool = False
    """Cache-coherent processing order per eval chunk (pays for lidar scans / random pixels, not for image patches)."""
    fused_training: bool = True
    """Training steps of a static scene on the fused nodes (models/neurad.py FusedTrainMixin); False: the reference's own
    get_nff_outputs over the HIP modules."""
    fused_losses: bool = True
    """Training steps: the lidar terms of get_metrics_dict (models/neurad.py:485-521) on csrc/losses.hip (one launch each way,
    the quantile as a radix select) and every per-step metric without boolean-mask indexing -- the reference's formulation
    costs ~15 `x[mask]` (a nonzero + a device->host read each) and a `float(beta)` per step, which serialise the host with the
    GPU.  Same keys, same values (tests/test_gpu_reference_plugin.py); False: the reference's own get_metrics_dict."""
    table_dtype: Literal["float32", "float16"] = "float32"
    """Storage of the main field's hash tables (static grid + actor grids).  "float16" (BASELINE config[4]; what tiny-cuda-nn
    stores): half the gather bytes; gradients arrive in fp16 and HashGridAdam keeps fp32 master copies (optim.py).  State
    dicts load into either (load_state_dict casts), so fp32 neurad checkpoints interchange."""
    proposal_table_dtype: Literal["float32", "float16"] = "float32"
    """... and of the proposal fields' static tables (their actor grids stay fp32: the fused sampler reads those as fp32)."""


class _NchwDecoderAdapter(torch.nn.Module):
    """what the reference's decode_features calls in place of ``self.rgb_decoder`` (models/neurad.py:362-365): it receives
    the NCHW VIEW of the pixel-major feature rows and hands an NCHW view of the pixel-major result back, so neither permute
    moves a byte, and the convolutions run on csrc/decoder.hip (model_components/cnns.py:decode_rgb)."""

    def __init__(self, decoder):
        super().__init__()
        self.__dict__["decoder"] = decoder  # not a registered submodule: the state_dict stays the reference's

    def forward(self, x):
        from neurad_studio_amd.model_components.cnns import decode_rgb

        b, c, h, w = x.shape
        return decode_rgb(self.decoder, x.permute(0, 2, 3, 1).reshape(-1, c).float(), (h, w)).permute(0, 3, 1, 2)


@contextlib.contextmanager
def _patched(module, name, value):
    old = getattr(module, name)
    setattr(module, name, value)
    try:
        yield
    finally:
        setattr(module, name, old)


class NeuRADHipModel(FusedEvalMixin, FusedTrainMixin, NeuRADModel):
    config: NeuRADHipModelConfig
    reference_output_keys = True  # get_metrics_dict consumes outputs["non_nearby_weights"] (models/neurad.py:508)

    def populate_modules(self):
        super().populate_modules()  # fields come out of config._target -> this package's classes
        cfg = self.config
        assert isinstance(self.field, HipNeuRADField), "config.field._target must be the HIP NeuRADField"
        self.fused_eval, self.early_stop_eps, self.order_rays = cfg.fused_eval, cfg.early_stop_eps, cfg.order_rays
        self.fused_training = cfg.fused_training
        s = cfg.sampling
        self.sampler = hip_samplers.ProposalNetworkSampler(
            num_proposal_samples_per_ray=s.num_proposal_samples, num_nerf_samples_per_ray=s.num_nerf_samples,
            num_proposal_network_iterations=cfg.num_proposal_rounds, single_jitter=s.single_jitter,
            initial_sampler=hip_samplers.PowerSampler(lambda_=s.power_lambda, scaling=s.power_scaling),
            update_sched=lambda x: 0)
        self.lidar_decoder = HipMLP(in_dim=cfg.field.nff_out_dim + cfg.appearance_dim, layer_width=32, out_dim=2,
                                    num_layers=3, out_activation=None)
        self.renderer_feat = hip_renderers.FeatureRenderer()
        self.renderer_accumulation = hip_renderers.AccumulationRenderer()
        self.renderer_depth = (hip_renderers.DepthRenderer(method="expected") if cfg.normalize_depth
                               else hip_renderers.render_depth_simple)
        self.interlevel_loss = hip_losses.zipnerf_interlevel_loss
        for what, name in ((cfg.table_dtype, "table_dtype"), (cfg.proposal_table_dtype, "proposal_table_dtype")):
            if what not in ("float32", "float16"):
                raise ValueError(f"{name} must be 'float32' or 'float16', got {what!r}")
        if cfg.table_dtype == "float16":
            for gr in [self.field.hashgrid.static_grid, *self.field.hashgrid.actor_grids]:
                gr.hash_table.data = gr.hash_table.data.half()
        if cfg.proposal_table_dtype == "float16":
            for pf in self.proposal_fields:
                pf.hashgrid.static_grid.hash_table.data = pf.hashgrid.static_grid.hash_table.data.half()

    def _render_weights(self, outputs, ray_samples):
        """models/neurad.py:711-724 without the cpu placeholder and independent of which ``nerfacc`` is importable"""
        with _patched(_ref_neurad, "nerfacc", hip_nerfacc):
            return super()._render_weights(outputs, ray_samples)

    def get_nff_outputs(self, ray_bundle, calc_lidar_losses: bool = False):
        if self.fused_eval_possible():
            return self.fused_nff_outputs(ray_bundle)
        if self.fused_training_possible():
            return self._fused_train_nff_outputs(ray_bundle, calc_lidar_losses)
        return super().get_nff_outputs(ray_bundle, calc_lidar_losses)

    _batch_layout = None

    def set_batch_layout(self, n_camera_rays, n_lidar_rays) -> None:
        """The caller vouches that the NEXT bundle is [n_camera_rays camera rays; n_lidar_rays lidar rays] in this order --
        what ``_merge_img_lidar`` builds (data/datamanagers/image_lidar_datamanager.py:379-423).  ``decode_features`` then
        splits the rendered features with two views instead of two boolean-mask gathers (each a nonzero + a host read).
        Set by integration/pipeline.py:ADHipPipeline, which owns the batch; consumed once."""
        self._batch_layout = None if n_camera_rays is None else (int(n_camera_rays), int(n_lidar_rays))

    def decode_features(self, features, patch_size, is_lidar=None, intensity_for_cam=False):
        """the reference's method (models/neurad.py:337-366) with the RGB CNN decoder on the HIP kernels: fp16 operands, fp32
        accumulation -- what the reference's mixed-precision trainer runs through MIOpen"""
        layout, self._batch_layout = self._batch_layout, None
        if (layout is not None and is_lidar is not None and not intensity_for_cam and features.is_cuda
                and sum(layout) == features.shape[0] and layout[0] > 0 and layout[1] > 0):
            # models/neurad.py:345-366 with the two masks replaced by the layout's views
            n_cam = layout[0]
            intensity, ray_drop_logit = self.lidar_decoder(features[n_cam:]).split(1, dim=-1)
            cam = features[:n_cam]
            if self.config.fused_decoder:
                from neurad_studio_amd.model_components.cnns import decode_rgb

                rgb = decode_rgb(self._modules["rgb_decoder"], cam.float(), tuple(patch_size))
            else:
                rgb = self.rgb_decoder(cam.view(-1, *patch_size, cam.shape[-1]).permute(0, 3, 1, 2)).permute(0, 2, 3, 1)
            return rgb, intensity.sigmoid(), ray_drop_logit
        if not (self.config.fused_decoder and features.is_cuda):
            return super().decode_features(features, patch_size, is_lidar, intensity_for_cam)
        decoder = self._modules["rgb_decoder"]
        self._modules["rgb_decoder"] = _NchwDecoderAdapter(decoder)
        try:
            return super().decode_features(features, patch_size, is_lidar, intensity_for_cam)
        finally:
            self._modules["rgb_decoder"] = decoder

    def get_metrics_dict(self, outputs, batch):
        if (self.training and self.config.fused_losses and "lidar" in batch and "image" in batch
                and outputs["depth"].is_cuda and "weights_list" in outputs):
            return self._fused_metrics_dict(outputs, batch)
        # the reference calls the module-level distortion_loss (models/neurad.py:524): the HIP one for this call only
        with _patched(_ref_neurad, "distortion_loss", hip_losses.distortion_loss):
            return super().get_metrics_dict(outputs, batch)

    def _fused_metrics_dict(self, outputs, batch):
        """get_metrics_dict (models/neurad.py:461-529) of a training step without a device->host read: same keys, same
        values.  The number of lidar rays comes with the batch (``batch["lidar"]`` has one row per lidar ray), so their
        positions are a compaction with a known size (ops.mask_compact) instead of a nonzero; the depth / intensity / ray-drop
        terms of the final samples and the depth terms of both proposal rounds are one launch each way (csrc/losses.hip); the
        four logging metrics are masked reductions."""
        from ..model_components.lidar_losses import LidarLossSettings, lidar_metrics, lidar_rows

        loss = self.config.loss
        m = {"psnr": self.psnr(outputs["rgb"].detach(), batch["image"])}
        is_lidar = batch["is_lidar"][:, 0]
        n_lidar = batch["lidar"].shape[0]
        rows = lidar_rows(is_lidar, n_lidar)
        did_return = batch["did_return"].index_select(0, rows[0])[:, 0]
        target_i, distance = batch["lidar"][..., 3:4], batch["distance"]
        with torch.no_grad():  # the four eval metrics the reference logs every step (models/neurad.py:477-483)
            ret = did_return[:, None]
            n_ret = ret.sum().clamp_min(1)
            pred = outputs["depth"].detach().index_select(0, rows[0])
            sq = (pred - distance) ** 2
            # median over the returned rays: the lower of the two middle values (torch.median), via a sort with the others at
            # +inf and a device-side index
            srt = torch.where(ret, sq, torch.full_like(sq, float("inf"))).reshape(-1).sort().values
            m["depth_median_l2"] = srt.gather(0, ((n_ret - 1) // 2).reshape(1))[0]
            m["depth_mean_rel_l2"] = (torch.where(ret, ((pred - distance) / distance) ** 2, torch.zeros_like(sq))).sum() / n_ret
            m["intensity_rmse"] = (torch.where(ret, (outputs["intensity"].detach() - target_i) ** 2,
                                               torch.zeros_like(sq)).sum() / n_ret).sqrt()
            m["ray_drop_accuracy"] = ((outputs["ray_drop_logits"].detach().sigmoid() > 0.5).squeeze(-1) == ~did_return).float().mean()
        cfg = LidarLossSettings(depth_mult=loss.depth_mult, intensity_mult=loss.intensity_mult, carving_mult=loss.carving_mult,
                                quantile_threshold=loss.quantile_threshold,
                                non_return_lidar_distance=loss.non_return_lidar_distance,
                                non_return_loss_mult=loss.non_return_loss_mult, ray_drop_loss_mult=loss.ray_drop_loss_mult,
                                prop_lidar_loss_mult=loss.prop_lidar_loss_mult)
        m.update(lidar_metrics(outputs, is_lidar, did_return, distance, target_i, cfg,
                               num_proposal_rounds=self.config.num_proposal_rounds, rows=rows))
        m["distortion"] = hip_losses.distortion_loss(outputs["weights_list"], outputs["ray_samples_list"])
        if self.config.field.use_sdf:
            m["sdf_to_density"] = self.field.sdf_to_density.beta.detach()  # (the reference: float(beta), a host read per step)
        self.camera_optimizer.get_metrics_dict(m)
        return m


def _trainer_config():
    # imported here, not at module level: nerfstudio.configs.method_configs runs the plugin discovery at its end
    # (plugins/registry.py:56-73), which imports THIS module -- at module level that is a cycle whenever this module is
    # imported before the method table
    from nerfstudio.configs.method_configs import method_configs

    cfg = deepcopy(method_configs["neurad"])  # schedules, data manager, the small groups' optimizers: the reference's own
    cfg.method_name = "neurad-hip"
    ref_model = cfg.pipeline.model
    cfg.pipeline.model = NeuRADHipModelConfig(eval_num_rays_per_chunk=ref_model.eval_num_rays_per_chunk,
                                              camera_optimizer=ref_model.camera_optimizer)
    # the pipeline: batches drawn on the device, gradients exchanged by GradientSynchronizer instead of DDP
    # (integration/pipeline.py); every field of the reference's pipeline / data manager config is carried over
    from .pipeline import ADHipDataManagerConfig, ADHipPipelineConfig

    ref_pipe, ref_dm = cfg.pipeline, cfg.pipeline.datamanager
    dm = ADHipDataManagerConfig(**{f.name: getattr(ref_dm, f.name) for f in dataclasses.fields(ref_dm)
                                   if f.name not in ("_target", "num_processes")})
    cfg.pipeline = ADHipPipelineConfig(**{f.name: getattr(ref_pipe, f.name) for f in dataclasses.fields(ref_pipe)
                                          if f.name not in ("_target", "datamanager")}, datamanager=dm)
    # the tables' optimizer on the HIP kernel, with the group's own hyper-parameters (configs/method_configs.py:423-426)
    ref_opt = cfg.optimizers["hashgrids"]["optimizer"]
    cfg.optimizers["hashgrids"]["optimizer"] = HashGridAdamConfig(lr=ref_opt.lr, eps=ref_opt.eps, max_norm=ref_opt.max_norm,
                                                                  weight_decay=ref_opt.weight_decay)
    # the small groups (MLPs, CNN decoder, trajectories, camera poses): the reference's optimizers and hyper-parameters, fused
    for name, group in cfg.optimizers.items():
        opt = group["optimizer"]
        if name != "hashgrids" and type(opt) in (AdamOptimizerConfig, AdamWOptimizerConfig):
            fused = FusedAdamWConfig if isinstance(opt, AdamWOptimizerConfig) else FusedAdamConfig
            group["optimizer"] = fused(lr=opt.lr, eps=opt.eps, max_norm=opt.max_norm, weight_decay=opt.weight_decay)
    # the trainer: the reference's iteration without its two grad_scaler.get_scale() host reads (integration/trainer.py)
    from .trainer import HipTrainerConfig

    return HipTrainerConfig(**{f.name: getattr(cfg, f.name) for f in dataclasses.fields(cfg) if f.name != "_target"})


_SPEC = None


def __getattr__(name):
    """``neurad_hip`` (the MethodSpecification the registry looks up) is built on first access"""
    global _SPEC
    if name != "neurad_hip":
        raise AttributeError(name)
    if _SPEC is None:
        spec = MethodSpecification(config=_trainer_config(),
                                   description="NeuRAD with the volumetric hot path on MI355X HIP kernels (libneurad_hip.so)")
        if _SPEC is None:  # (building it may have run the discovery, which built it already)
            _SPEC = spec
    return _SPEC
