"""The hot path of NeuRADModel: ``get_nff_outputs`` and what it calls
(mirror of nerfstudio/models/neurad.py:96-117,226-254,368-459,677-734).

Decoders, losses, metrics, camera optimisation and data loading stay in neurad-studio (out of scope);
this module owns: far clamp + sampler + sky stretch (M1), field (F1), weights (C1), compositing (C2),
appearance embedding (C3), proposal outputs (C4).

eval / no-grad -> 2 kernels per ray batch: fused proposal sampler + fused field/compositing.
training        -> reference orchestration over operator-level HIP ops with hand-written backward."""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch
from torch import Tensor, nn

from ..cameras.rays import RayBundle, RaySamples
from ..field_components.field_heads import FieldHeadNames
from ..fields.neurad_field import NeuRADField, NeuRADFieldConfig, NeuRADProposalField, NeuRADProposalFieldConfig
from ..model_components.ray_samplers import PowerSampler, ProposalNetworkSampler
from ..model_components.renderers import AccumulationRenderer, DepthRenderer, FeatureRenderer, render_depth_simple
from ..shims import nerfacc

EPS = 1e-7


@dataclass
class SamplingSettings:  # models/neurad.py:96-117
    single_jitter: bool = True
    proposal_field_1: NeuRADProposalFieldConfig = field(default_factory=NeuRADProposalFieldConfig)
    proposal_field_2: NeuRADProposalFieldConfig = field(default_factory=NeuRADProposalFieldConfig)
    num_proposal_samples: Tuple[int, ...] = (128, 64)
    num_nerf_samples: int = 32
    power_lambda: float = -1.0
    power_scaling: float = 0.1
    sky_distance: float = 20000.0


@dataclass
class NeuRADHotPathConfig:
    sampling: SamplingSettings = field(default_factory=SamplingSettings)
    field: NeuRADFieldConfig = field(default_factory=NeuRADFieldConfig)
    appearance_dim: int = 16
    use_temporal_appearance: bool = True
    temporal_appearance_freq: float = 1.0
    rgb_upsample_factor: int = 3
    normalize_depth: bool = False
    carving_epsilon: float = 0.1
    non_return_lidar_distance: float = 150.0
    reproduce_late_binding_quirk: bool = True
    """models/neurad.py:248 builds density_fns with a late-binding closure, so BOTH proposal rounds evaluate
    proposal_fields[1].  True keeps parity with the reference; False uses field i in round i."""


class NeuRADHotPath(nn.Module):
    def __init__(self, config: NeuRADHotPathConfig, static_scale: float, num_sensors: int = 1, duration: float = 1.0,
                 actors=None) -> None:
        super().__init__()
        self.config = config
        self.field = config.field.setup(actors=actors, static_scale=static_scale)
        self._duration = duration
        if config.appearance_dim > 0:
            self._num_embeds_per_sensor = (math.ceil(duration * config.temporal_appearance_freq)
                                           if config.use_temporal_appearance else 1)
            self.appearance_embedding = nn.Embedding(num_sensors * self._num_embeds_per_sensor, config.appearance_dim)
        s = config.sampling
        self.sampler = ProposalNetworkSampler(
            num_proposal_samples_per_ray=s.num_proposal_samples, num_nerf_samples_per_ray=s.num_nerf_samples,
            num_proposal_network_iterations=len(s.num_proposal_samples), single_jitter=s.single_jitter,
            initial_sampler=PowerSampler(lambda_=s.power_lambda, scaling=s.power_scaling), update_sched=lambda x: 0)
        self.proposal_fields = nn.ModuleList([c.setup(actors=actors, static_scale=static_scale)
                                              for c in (s.proposal_field_1, s.proposal_field_2)])
        if config.reproduce_late_binding_quirk:
            last = self.proposal_fields[-1]
            self.density_fns = [lambda x, f=last: f.get_density(x)[0] for _ in self.proposal_fields]
        else:
            self.density_fns = [lambda x, f=f: f.get_density(x)[0] for f in self.proposal_fields]
        self.renderer_feat = FeatureRenderer()
        self.renderer_accumulation = AccumulationRenderer()
        self.renderer_depth = DepthRenderer(method="expected") if config.normalize_depth else render_depth_simple

    @property
    def fields(self):
        return [self.field, *self.proposal_fields]

    def get_param_groups(self) -> Dict[str, List[nn.Parameter]]:
        groups: Dict[str, List[nn.Parameter]] = {"hashgrids": [], "fields": []}
        for f in self.fields:
            f.get_param_groups(groups)
        if self.config.appearance_dim > 0:
            groups["fields"] += list(self.appearance_embedding.parameters())
        return groups

    # ---- M1 (models/neurad.py:443-459) ------------------------------------------------------------
    def _prepare_bundle(self, ray_bundle: RayBundle) -> float:
        sky = self.config.sampling.sky_distance
        if ray_bundle.fars is not None:
            ray_bundle.fars.clamp_max_(sky)
        else:
            ray_bundle.fars = torch.full_like(ray_bundle.pixel_area, sky)
        if ray_bundle.nears is None:
            ray_bundle.nears = torch.zeros_like(ray_bundle.fars)
        return sky

    @staticmethod
    def _stretch_sky(ray_samples: RaySamples, sky: float) -> None:
        dist_to_sky = sky - ray_samples.frustums.ends[..., -1, 0]
        ray_samples.frustums.ends[..., -1, 0] += dist_to_sky
        ray_samples.deltas[..., -1, 0] += dist_to_sky
        ray_samples.spacing_ends[..., -1, 0] = 1 - EPS

    def _get_ray_samples(self, ray_bundle: RayBundle):
        sky = self._prepare_bundle(ray_bundle)
        if torch.is_grad_enabled() or self.training or self.field.hashgrid.has_actors():
            ray_samples, prop_weights, prop_ray_samples = self.sampler(ray_bundle, self.density_fns, pass_ray_samples=True)
            # bins come out of the kernels as views of one [R,S+1] edge tensor: materialise before the in-place stretch
            fr = ray_samples.frustums
            fr.ends, ray_samples.deltas = fr.ends.clone(), ray_samples.deltas.clone()
            ray_samples.spacing_ends = ray_samples.spacing_ends.clone()
        else:
            pf = list(self.proposal_fields)
            if self.config.reproduce_late_binding_quirk:
                pf = [pf[-1]] * len(pf)
            ray_samples, prop_weights, prop_ray_samples = self.sampler.generate_fused(ray_bundle, pf, sky)
            fr = ray_samples.frustums
            fr.ends, ray_samples.deltas = fr.ends.clone(), ray_samples.deltas.clone()
            ray_samples.spacing_ends = ray_samples.spacing_ends.clone()
        self._stretch_sky(ray_samples, sky)
        if self.training and "is_lidar" in ray_bundle.metadata:
            self._compute_is_close_to_lidar(ray_samples, *prop_ray_samples)
        return ray_samples, prop_ray_samples, prop_weights

    def _scale_pixel_area(self, ray_bundle: RayBundle):  # models/neurad.py:702-709
        is_lidar = ray_bundle.metadata.get("is_lidar")
        if is_lidar is not None:
            scaling = torch.ones_like(ray_bundle.pixel_area)
            scaling[~is_lidar] = self.config.rgb_upsample_factor**2
        else:
            scaling = self.config.rgb_upsample_factor**2
        ray_bundle.pixel_area = ray_bundle.pixel_area * scaling

    def _render_weights(self, outputs, ray_samples):  # models/neurad.py:711-724 (no cpu placeholder: GPU only)
        if self.config.field.use_sdf:
            weights, _ = nerfacc.render_weight_from_alpha(outputs[FieldHeadNames.ALPHA].squeeze(-1))
        else:
            weights, _, _ = nerfacc.render_weight_from_density(
                t_ends=ray_samples.frustums.ends.squeeze(-1), t_starts=ray_samples.frustums.starts.squeeze(-1),
                sigmas=outputs[FieldHeadNames.DENSITY].squeeze(-1))
        return weights

    def _get_appearance_embedding(self, ray_bundle, features):  # models/neurad.py:423-441
        sensor_idx = ray_bundle.metadata.get("sensor_idxs")
        if sensor_idx is None:
            assert not self.training, "sensor_idxs must be present in metadata during training"
            sensor_idx = torch.zeros_like(features[..., :1], dtype=torch.long)
        if self.config.use_temporal_appearance:
            n = self._num_embeds_per_sensor
            time_idx = ray_bundle.times / self._duration * n
            before = time_idx.floor().clamp(0, n - 1)
            after = (before + 1).clamp(0, n - 1)
            ratio = time_idx - before
            before, after = (x + sensor_idx * n for x in (before, after))
            be = self.appearance_embedding(before.squeeze(-1).long())
            ae = self.appearance_embedding(after.squeeze(-1).long())
            return be * (1 - ratio) + ae * ratio
        return self.appearance_embedding(sensor_idx.squeeze(-1))

    def _compute_is_close_to_lidar(self, *all_ray_samples):  # models/neurad.py:677-700
        for rs in all_ray_samples:
            if rs is None:
                continue
            md, fr = rs.metadata, rs.frustums
            sample_distance = (fr.starts + fr.ends) * 0.5
            mask = md["is_lidar"].clone()
            idx = mask.nonzero(as_tuple=True)
            sd = sample_distance[idx]
            dist = md["directions_norm"][idx] - sd
            close = dist.abs() < self.config.carving_epsilon
            if "did_return" in md:
                did_return = md["did_return"][idx]
                mask[idx] = (did_return & close) | ((~did_return) & (sd < self.config.non_return_lidar_distance))
            else:
                mask[idx] = close
            md["is_close_to_lidar"] = mask

    # ---- get_nff_outputs (models/neurad.py:368-421) ------------------------------------------------
    def get_nff_outputs(self, ray_bundle: RayBundle, calc_lidar_losses: bool = False) -> Dict[str, Tensor]:
        self._scale_pixel_area(ray_bundle)
        ray_samples, proposal_ray_samples, proposal_weights = self._get_ray_samples(ray_bundle)
        fused = not (torch.is_grad_enabled() or self.training) and self.field.fused_supported()
        fr = ray_samples.frustums
        if fused:
            feats, depth, accumulation = self.field.render(ray_bundle.origins, ray_bundle.directions,
                                                           ray_bundle.pixel_area, fr.starts[..., 0], fr.ends[..., 0])
            if self.config.normalize_depth:
                raise NotImplementedError("normalize_depth with the fused kernel")
            features, weights = feats, None
        else:
            outputs = self.field(ray_samples)
            weights = self._render_weights(outputs, ray_samples)
            accumulation = self.renderer_accumulation(weights=weights[..., None])
            weights = torch.cat((weights[..., :-1], weights[..., -1:] + 1 - accumulation), dim=-1).unsqueeze(-1)
            features = self.renderer_feat(features=outputs[FieldHeadNames.FEATURE], weights=weights)
            weights, ray_samples = weights[..., :-1, :], ray_samples[..., :-1]
            depth = self.renderer_depth(weights=weights, ray_samples=ray_samples)
        if self.config.appearance_dim > 0:
            features = torch.cat([features, self._get_appearance_embedding(ray_bundle, features)], dim=-1)
        nff = {"features": features, "depth": depth, "accumulation": accumulation}
        for i, (pw, prs) in enumerate(zip(proposal_weights, proposal_ray_samples)):
            nff[f"prop_depth_{i}"] = self.renderer_depth(pw, prs)
            if self.training and calc_lidar_losses:
                m = (~prs.metadata["is_close_to_lidar"]) & prs.metadata["is_lidar"]
                nff[f"prop_weights_loss_{i}"] = ((pw * m) ** 2).sum()
        if self.training:
            nff["weights_list"] = proposal_weights + [weights]
            nff["ray_samples_list"] = proposal_ray_samples + [ray_samples]
        return nff

    def forward(self, ray_bundle: RayBundle, calc_lidar_losses: bool = False):
        return self.get_nff_outputs(ray_bundle, calc_lidar_losses)
