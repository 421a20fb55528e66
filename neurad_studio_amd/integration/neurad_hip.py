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
from nerfstudio.engine.optimizers import AdamOptimizerConfig
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

    def decode_features(self, features, patch_size, is_lidar=None, intensity_for_cam=False):
        """the reference's method (models/neurad.py:337-366) with the RGB CNN decoder on the HIP kernels: fp16 operands, fp32
        accumulation -- what the reference's mixed-precision trainer runs through MIOpen"""
        if not (self.config.fused_decoder and features.is_cuda):
            return super().decode_features(features, patch_size, is_lidar, intensity_for_cam)
        decoder = self._modules["rgb_decoder"]
        self._modules["rgb_decoder"] = _NchwDecoderAdapter(decoder)
        try:
            return super().decode_features(features, patch_size, is_lidar, intensity_for_cam)
        finally:
            self._modules["rgb_decoder"] = decoder

    def get_metrics_dict(self, outputs, batch):
        # the reference calls the module-level distortion_loss (models/neurad.py:524): the HIP one for this call only
        with _patched(_ref_neurad, "distortion_loss", hip_losses.distortion_loss):
            return super().get_metrics_dict(outputs, batch)


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
    # the tables' optimizer on the HIP kernel, with the group's own hyper-parameters (configs/method_configs.py:423-426)
    ref_opt = cfg.optimizers["hashgrids"]["optimizer"]
    cfg.optimizers["hashgrids"]["optimizer"] = HashGridAdamConfig(lr=ref_opt.lr, eps=ref_opt.eps, max_norm=ref_opt.max_norm,
                                                                  weight_decay=ref_opt.weight_decay)
    return cfg


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
