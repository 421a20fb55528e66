"""The hot path of NeuRADModel: ``get_nff_outputs`` and what it calls
(nerfstudio/models/neurad.py:96-117,226-254,368-459,677-734).

Two users:
  * ``FusedEvalMixin`` -- mixed into the nerfstudio plugin model (integration/neurad_hip.py: a subclass of the
    reference's own NeuRADModel) so that eval chunks run as two kernels (fused proposal sampler + fused
    field/compositing) while training keeps the reference's own ``get_nff_outputs`` over this package's fields,
    sampler and renderers;
  * ``NeuRADHotPath`` -- the same path as a standalone module (no nerfstudio import), used by the GPU tests, the
    benchmarks and by anyone who wants the path without the trainer.  Decoders, image losses, metrics, camera
    optimisation and data loading stay in neurad-studio.

eval / no-grad -> 2 kernels per ray batch (+ the optional ray-ordering pass);
training       -> operator-level HIP ops with hand-written backward (fields: one autograd node each)."""
from __future__ import annotations

import contextlib
import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch
from torch import Tensor, nn

from .. import autograd as ag
from .. import ops
from ..cameras.rays import Frustums, RayBundle, RaySamples, sample_times
from ..field_components.field_heads import FieldHeadNames
from ..fields.neurad_field import NeuRADField, NeuRADFieldConfig, NeuRADProposalField, NeuRADProposalFieldConfig
from ..model_components.ray_samplers import PowerSampler, PowerSpacing, ProposalNetworkSampler
from ..model_components.renderers import AccumulationRenderer, DepthRenderer, FeatureRenderer, render_depth_simple
from ..shims import nerfacc

EPS = 1e-7


class FusedEvalMixin:
    """Eval-time ``get_nff_outputs`` on the fused kernels.  Expects on ``self`` what NeuRADModel.populate_modules builds
    (models/neurad.py:167-254): ``config`` (sampling.*, field.use_sdf, appearance_dim, normalize_depth), ``field``,
    ``proposal_fields``, ``sampler`` (this package's ProposalNetworkSampler), ``renderer_depth``,
    ``_scale_pixel_area`` and ``_get_appearance_embedding``."""

    fused_eval: bool = True
    """False forces the operator-level path in eval too (A/B, debugging)."""
    early_stop_eps: float = 0.0
    """> 0: eval rays stop marching once their transmittance is below it (error bounded by it; 0 = exact)."""
    order_rays: bool = False
    """Compute a cache-coherent processing order per eval chunk (ops.ray_order).  Pays for incoherent batches (lidar
    scans, random pixels); camera patches are coherent as they come."""
    reproduce_late_binding_quirk: bool = True
    """models/neurad.py:248 builds density_fns with a late-binding closure, so BOTH proposal rounds evaluate
    proposal_fields[1]; the fused sampler has to be told the same."""

    def fused_eval_possible(self) -> bool:
        return (self.fused_eval and not torch.is_grad_enabled() and not self.training
                and self.field.fused_supported(with_actors=True))

    def fused_nff_outputs(self, ray_bundle) -> Dict[str, Tensor]:
        cfg = self.config
        sky = cfg.sampling.sky_distance
        self._scale_pixel_area(ray_bundle)
        if ray_bundle.fars is not None:  # M1: far clamp, near default (models/neurad.py:443-449)
            ray_bundle.fars.clamp_max_(sky)
        else:
            ray_bundle.fars = torch.full_like(ray_bundle.pixel_area, sky)
        if ray_bundle.nears is None:
            ray_bundle.nears = torch.zeros_like(ray_bundle.fars)
        o, d = ray_bundle.origins.contiguous(), ray_bundle.directions.contiguous()
        pf = list(self.proposal_fields)
        if self.reproduce_late_binding_quirk:
            pf = [pf[-1]] * len(pf)
        times, cand = None, None
        field_actors = self.field.hashgrid.has_actors()
        prop_actors = [f.hashgrid.has_actors() for f in pf]
        if any(prop_actors) and not (all(prop_actors) and all(f.fused_sampler_supported() for f in pf)):
            # actor grids the fused sampler is not instantiated for (or actors in only some of the proposal fields): the
            # proposal rounds run as operator-level kernels (static density + the actor overlay); the field + compositing
            # stay one kernel with per-sample table select
            ray_samples, prop_ray_samples, prop_weights = self._get_ray_samples(ray_bundle)
            fr = ray_samples.frustums
            starts, ends = fr.starts[..., 0].contiguous(), fr.ends[..., 0].contiguous()
            times = sample_times(ray_samples)
        else:
            if field_actors or any(prop_actors):
                # Which actors a ray can meet depends on its LINE only (bounding-sphere cull, neurad_encoding.py:225-247):
                # one candidate list per ray, from any two samples on it, serves both proposal rounds and the field.
                if ray_bundle.times is None:
                    raise ValueError("dynamic actors need ray times")
                times = ray_bundle.times.reshape(-1)
                n = ray_bundle.nears.reshape(-1)  # two one-metre samples at the start of the ray: a well-conditioned line
                hg = self.field.hashgrid if field_actors else pf[0].hashgrid
                _, cand = hg.prepare_actors(o, d, ray_bundle.pixel_area.reshape(-1), torch.stack([n, n + 1], -1),
                                            torch.stack([n + 1, n + 2], -1), times)
            ray_samples, prop_weights, prop_ray_samples = self.sampler.generate_fused(
                ray_bundle, pf, sky, actor_cand=cand if any(prop_actors) else None)
            fr = ray_samples.frustums
            starts = fr.starts[..., 0]
            ends = fr.ends[..., 0].clone()
            ends[:, -1] = sky  # the sky stretch of the last sample (models/neurad.py:451-455)
        order = ops.ray_order(o, d, self.field.hashgrid.static_scale) if self.order_rays else None
        want_w = bool(cfg.normalize_depth)
        out = self.field.render(o, d, ray_bundle.pixel_area, starts, ends, return_weights=want_w,
                                early_stop_eps=self.early_stop_eps, order=order, times=times, actor_cand=cand)
        features, depth, accumulation = out[:3]
        if want_w:  # DepthRenderer("expected") over the non-sky samples (renderers.py:398-416)
            w = out[3][:, :-1]
            steps = (starts[:, :-1] + ends[:, :-1]) / 2
            depth = (depth / (w.sum(-1, keepdim=True) + 1e-10)).clip(steps.min(), steps.max())
        if cfg.appearance_dim > 0:
            features = torch.cat([features, self._get_appearance_embedding(ray_bundle, features)], dim=-1)
        nff = {"features": features, "depth": depth, "accumulation": accumulation}
        for i, (pw, prs) in enumerate(zip(prop_weights, prop_ray_samples)):
            nff[f"prop_depth_{i}"] = self.renderer_depth(pw, prs)
        return nff


class FusedTrainMixin:
    """Training-time ``get_nff_outputs`` as a handful of autograd nodes (autograd.ProposalRoundFn per
    sampler round, autograd.NffRenderTrainFn for field + head + compositing + appearance) instead of the reference's
    orchestration over RaySamples views -- same outputs, ~1/4 of the launches.  Mixed into ``NeuRADHotPath`` and into the
    nerfstudio plugin model (integration/neurad_hip.py: a subclass of the reference's NeuRADModel); expects on ``self`` what
    ``NeuRADModel.populate_modules`` builds (models/neurad.py:167-254)."""

    fused_training: bool = True
    """False keeps the operator-level path (the reference's own orchestration over this package's modules): A/B, debugging."""
    reference_output_keys: bool = False
    """True (the plugin): also return ``non_nearby_weights`` as the reference's get_metrics_dict consumes it
    (models/neurad.py:508: only its squared sum is used) -- here the DENSE masked weights, same squared sum, no nonzero /
    host sync."""

    def _carving_cfg(self):
        c = getattr(self.config, "loss", self.config)  # the reference keeps these under config.loss (models/neurad.py:79,87)
        return c.carving_epsilon, c.non_return_lidar_distance

    def _scale_pixel_area_nosync(self, ray_bundle) -> None:
        """_scale_pixel_area (models/neurad.py:702-709) as one ``where``: the reference's masked assignment is an index_put
        with a boolean mask, i.e. a nonzero + a device->host read per step"""
        up2 = float(self.config.rgb_upsample_factor**2)
        is_lidar = ray_bundle.metadata.get("is_lidar")
        if is_lidar is None:
            ray_bundle.pixel_area = ray_bundle.pixel_area * up2
        else:
            ray_bundle.pixel_area = torch.where(is_lidar, ray_bundle.pixel_area, ray_bundle.pixel_area * up2)

    def fused_training_possible(self) -> bool:
        f = self.field
        return (self.fused_training and self.training and torch.is_grad_enabled() and f.fused_training
                and f.fused_supported(with_actors=True) and f._fused_train_ok()
                and isinstance(self.sampler.initial_sampler, PowerSampler))

    def _fused_train_nff_outputs(self, ray_bundle: RayBundle, calc_lidar_losses: bool) -> Dict[str, Tensor]:
        """get_nff_outputs (models/neurad.py:368-421) without the RaySamples plumbing: bin edges [R,S+1] go from kernel to
        kernel.  ``ray_samples_list`` holds light RaySamples (edges as views, no deltas / metadata: the carving terms
        recompute their masks from the edges)."""
        cfg, smp = self.config, self.sampler
        sky = cfg.sampling.sky_distance
        self._scale_pixel_area_nosync(ray_bundle)
        if ray_bundle.fars is None:
            ray_bundle.fars = torch.full_like(ray_bundle.pixel_area, sky)
        else:
            ray_bundle.fars.clamp_max_(sky)
        if ray_bundle.nears is None:
            ray_bundle.nears = torch.zeros_like(ray_bundle.fars)
        o, d = ray_bundle.origins.contiguous(), ray_bundle.directions.contiguous()
        a = ray_bundle.pixel_area.reshape(-1)
        R, dev = o.shape[0], o.device
        init, pdf = smp.initial_sampler, smp.pdf_sampler
        rounds = smp.num_proposal_network_iterations
        counts = tuple(smp.num_proposal_samples_per_ray[:rounds]) + (smp.num_nerf_samples_per_ray,)
        fn = PowerSpacing(ray_bundle.nears, ray_bundle.fars, init.lambda_, init.scaling)
        t_rand = None
        if init.train_stratified and init.training:
            t_rand = (torch.rand((R, 1), device=dev).expand(R, counts[0] + 1).contiguous() if init.single_jitter
                      else torch.rand((R, counts[0] + 1), device=dev))
        sp, eu = ops.power_sampler(ray_bundle.nears, ray_bundle.fars, counts[0], init.lambda_, init.scaling, t_rand)
        pfs = list(self.proposal_fields)
        if getattr(self, "reproduce_late_binding_quirk", True):
            pfs = [pfs[-1]] * len(pfs)
        train_props = smp._proposals_train_this_step()
        cand = None
        if self.field.hashgrid.has_actors() or any(p.hashgrid.has_actors() for p in pfs):
            # which actors a ray can meet depends on its LINE only (bounding-sphere cull, neurad_encoding.py:225-247): one
            # candidate list per step serves both proposal rounds and the field, as in eval (was: one launch per field call)
            if ray_bundle.times is None:
                raise ValueError("dynamic actors need ray times")
            n0 = ray_bundle.nears.reshape(-1)
            hg0 = self.field.hashgrid if self.field.hashgrid.has_actors() else next(p.hashgrid for p in pfs if p.hashgrid.has_actors())
            with torch.no_grad():
                _, cand = hg0.prepare_actors(o, d, a, torch.stack([n0, n0 + 1], -1), torch.stack([n0 + 1, n0 + 2], -1),
                                             ray_bundle.times.reshape(-1))
        nff: Dict[str, Tensor] = {}
        weights_list, samples_list = [], []
        lidar_terms = self.training and calc_lidar_losses
        if lidar_terms:
            md = ray_bundle.metadata
            carve = (md["is_lidar"], md.get("did_return"), md["directions_norm"], *self._carving_cfg())
        for k in range(rounds):
            pf = pfs[k]
            g = pf.hashgrid.static_grid
            with contextlib.nullcontext() if train_props else torch.no_grad():  # frozen between scheduled updates
                if pf.hashgrid.has_actors():
                    # dynamic actors: the field's own density (static kernel + the actor overlay, operator level), then
                    # weights and the round's depth from the edges
                    dens = pf.get_density(_light_samples(ray_bundle, sp, eu, fn), actor_cand=cand)[0][..., 0]
                    w, pdepth = ag.PropWeightsFn.apply(eu, dens.contiguous())
                else:
                    w, pdepth = ag.ProposalRoundFn.apply(g.hash_table, pf.density_decoder.weight, g.spec,
                                                         pf.hashgrid.static_scale, o, d, a, eu)
            weights_list.append(w[..., None])
            samples_list.append(_light_samples(ray_bundle, sp, eu, fn))
            nff[f"prop_depth_{k}"] = _expected_depth(pdepth, w, eu) if cfg.normalize_depth else pdepth
            if lidar_terms:
                nff[f"prop_weights_loss_{k}"] = ag.CarvingLossFn.apply(w, eu[:, :-1], eu[:, 1:], *carve)
            rand = None
            if pdf.train_stratified and pdf.training:
                rand = torch.rand((R,) if pdf.single_jitter else (R, counts[k + 1] + 1), device=dev)
            wd = w.detach()
            sp, eu = ops.pdf_sample(wd if smp._anneal == 1.0 else wd.pow(smp._anneal), sp, fn.nears, fn.fars, counts[k + 1],
                                    fn.lam, fn.scaling, pdf.histogram_padding, rand)
        if train_props:
            smp._steps_since_update = 0
        eu[:, -1] = sky  # the sky stretch of the last sample (models/neurad.py:451-455)
        sp[:, -1] = 1 - EPS
        appearance = None
        if cfg.appearance_dim > 0:
            sensor = ray_bundle.metadata.get("sensor_idxs")
            assert sensor is not None, "sensor_idxs must be present in metadata during training"
            appearance = (self.appearance_embedding.weight, sensor, ray_bundle.times if cfg.use_temporal_appearance else None,
                          (float(self._duration), int(self._num_embeds_per_sensor), bool(cfg.use_temporal_appearance)))
        features, depth, accumulation, w_ns = self.field.render_train(
            o, d, a, eu, appearance, times=ray_bundle.times, actor_cand=cand if self.field.hashgrid.has_actors() else None)
        S = counts[-1]
        if cfg.normalize_depth:  # over the non-sky samples (models/neurad.py:386-391)
            depth = _expected_depth(depth, w_ns, eu[:, :S])
        nff.update(features=features, depth=depth, accumulation=accumulation)
        if self.training:
            nff["weights_list"] = weights_list + [w_ns[..., None]]
            # the sky sample plays no further role: the first S-1 samples = the first S edges
            nff["ray_samples_list"] = samples_list + [_light_samples(ray_bundle, sp[:, :S], eu[:, :S], fn)]
        if lidar_terms:
            # sum((w * (is_lidar & ~is_close))^2) of the final samples: what the reference forms from `non_nearby_weights`
            # (models/neurad.py:410-419,508-509) -- selecting them needs a nonzero + a host sync
            if self.reference_output_keys:
                close, _, _ = ops.lidar_carving(eu[:, :S - 1], eu[:, 1:S], *carve, want_mask=True)
                nff["non_nearby_weights"] = (w_ns * (carve[0].reshape(-1, 1) & ~close))[..., None]
                nff["non_nearby_lidar_ray_indices"] = None  # (unused by the reference: models/neurad.py:507)
            else:
                nff["non_nearby_weights_loss"] = ag.CarvingLossFn.apply(w_ns, eu[:, :S - 1], eu[:, 1:S], *carve)
        return nff


def _expected_depth(depth: Tensor, w: Tensor, edges: Tensor) -> Tensor:
    """DepthRenderer("expected") (model_components/renderers.py:398-416) from what the fused nodes return: depth = sum w * mid
    [R,1], the weights w [R,S] of those samples and their bin edges [R,S+1] -> sum w * mid / (sum w + 1e-10), clipped to the
    batch's range of sample midpoints"""
    mid = (edges[:, :-1] + edges[:, 1:]) / 2
    return torch.clip(depth / (w.sum(-1, keepdim=True) + 1e-10), mid.min(), mid.max())


def _light_samples(rb: RayBundle, sp: Tensor, eu: Tensor, fn) -> RaySamples:
    """RaySamples of the fused training path: S samples from S+1 edges, every field a VIEW (per-ray fields stride-0 like
    rays.py:336-355, starts / ends two views of the edge tensor); no deltas, no metadata -- nothing is launched"""
    S = eu.shape[1] - 1

    def ex(t):
        return None if t is None else t[..., None, :].expand(*t.shape[:-1], S, t.shape[-1])

    fr = Frustums(ex(rb.origins), ex(rb.directions), eu[:, :-1, None], eu[:, 1:, None], ex(rb.pixel_area))
    return RaySamples(frustums=fr, spacing_starts=sp[:, :-1, None], spacing_ends=sp[:, 1:, None],
                      spacing_to_euclidean_fn=fn, times=ex(rb.times), sdist=sp)


def _slice_bundle(rb, a: int, b: int):
    """rays [a, b) of a flat bundle: the reference's RayBundle slices itself (rays.py:293-311), this package's is a plain
    dataclass of [R, C] tensors"""
    if hasattr(rb, "get_row_major_sliced_ray_bundle"):
        return rb.get_row_major_sliced_ray_bundle(a, b)

    def sl(t):
        return None if t is None else t[a:b]

    return RayBundle(origins=sl(rb.origins), directions=sl(rb.directions), pixel_area=sl(rb.pixel_area),
                     camera_indices=sl(rb.camera_indices), nears=sl(rb.nears), fars=sl(rb.fars),
                     metadata={k: sl(v) for k, v in rb.metadata.items()}, times=sl(rb.times),
                     termination_distances=sl(rb.termination_distances))


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
    """see FusedEvalMixin; False uses field i in round i."""
    lidar_decoder: bool = True
    """Build the lidar head (intensity, ray-drop logit): MLP 48 -> 32 -> 32 -> 2 (models/neurad.py:217-224), the first
    consumer of the rendered features (SURVEY §8(f) row 1); the RGB CNN decoder stays in neurad-studio."""


class NeuRADHotPath(FusedEvalMixin, FusedTrainMixin, nn.Module):
    def __init__(self, config: NeuRADHotPathConfig, static_scale: float, num_sensors: int = 1, duration: float = 1.0,
                 actors=None) -> None:
        super().__init__()
        self.config = config
        self.reproduce_late_binding_quirk = config.reproduce_late_binding_quirk
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
        if config.lidar_decoder:
            from ..field_components.mlp import MLP

            self.lidar_decoder = MLP(in_dim=config.field.nff_out_dim + config.appearance_dim, num_layers=3,
                                     layer_width=32, out_dim=2)

    @property
    def fields(self):
        return [self.field, *self.proposal_fields]

    def get_param_groups(self) -> Dict[str, List[nn.Parameter]]:
        groups: Dict[str, List[nn.Parameter]] = {"hashgrids": [], "fields": []}
        for f in self.fields:
            f.get_param_groups(groups)
        if self.config.lidar_decoder:
            groups["fields"] += list(self.lidar_decoder.parameters())
        if self.config.appearance_dim > 0:
            groups["fields"] += list(self.appearance_embedding.parameters())
        return groups

    def decode_lidar(self, features: Tensor, is_lidar: Optional[Tensor] = None, rows: Optional[Tensor] = None
                     ) -> Tuple[Tensor, Tensor]:
        """lidar head of decode_features (models/neurad.py:341-348): rendered features of the lidar rays ->
        (intensity in [0,1], ray-drop logit), each [n_lidar, 1].  rows (int64 [n_lidar], lidar_losses.lidar_rows): the
        lidar rays' positions in the batch -- `features[is_lidar]` costs a nonzero and a host sync for the same thing"""
        if rows is not None:
            features = features.index_select(0, rows)
        elif is_lidar is not None:
            features = features[is_lidar.reshape(-1)]
        intensity, ray_drop_logit = self.lidar_decoder(features).split(1, dim=-1)
        return intensity.sigmoid(), ray_drop_logit

    # ---- chunked eval entry (models/neurad.py:623-675) ----------------------------------------------
    @torch.no_grad()
    def get_outputs_for_ray_bundle(self, ray_bundle, num_rays_per_chunk: int = 1 << 17,
                                   is_lidar: bool = False) -> Dict[str, Tensor]:
        """Render a whole image / lidar scan: flat bundle of N rays -> {features, depth, accumulation, prop_depth_i}
        [N, C], plus the lidar head's outputs (intensity, ray_drop_logits, ray_drop_prob) for a lidar scan.  The
        reference walks 2^15-ray chunks and concatenates (``get_outputs_for_camera_ray_bundle``); here the chunks are
        large (the activations of 2^17 rays are a few hundred MB of 288 GB) and each writes its slice of buffers
        allocated once.  Camera features go to neurad-studio's CNN decoder (decode_features, models/neurad.py:328-366)."""
        n = len(ray_bundle)
        out: Dict[str, Tensor] = {}
        for a in range(0, n, num_rays_per_chunk):
            b = min(a + num_rays_per_chunk, n)
            chunk = self.get_nff_outputs(_slice_bundle(ray_bundle, a, b))
            for k, v in chunk.items():
                if k not in out:
                    out[k] = torch.empty((n, *v.shape[1:]), dtype=v.dtype, device=v.device)
                out[k][a:b] = v
        if is_lidar and self.lidar_decoder is not None:
            intensity, logit = self.decode_lidar(out["features"])
            out["intensity"], out["ray_drop_logits"], out["ray_drop_prob"] = intensity, logit, logit.sigmoid()
        return out

    # ---- M1 (models/neurad.py:443-459), operator-level path ---------------------------------------
    def _get_ray_samples(self, ray_bundle: RayBundle):
        sky = self.config.sampling.sky_distance
        if ray_bundle.fars is None:
            ray_bundle.fars = torch.full_like(ray_bundle.pixel_area, sky)
        else:
            ray_bundle.fars.clamp_max_(sky)
        if ray_bundle.nears is None:
            ray_bundle.nears = torch.zeros_like(ray_bundle.fars)
        ray_samples, prop_weights, prop_ray_samples = self.sampler(ray_bundle, self.density_fns, pass_ray_samples=True)
        # sky stretch, in place on the sampler's edge tensors: the last bin ends at sky_distance
        fr = ray_samples.frustums
        stretch = sky - fr.ends[..., -1, 0]
        fr.ends[..., -1, 0] += stretch
        ray_samples.deltas[..., -1, 0] += stretch
        ray_samples.spacing_ends[..., -1, 0] = 1 - EPS
        if self.training and "is_lidar" in ray_bundle.metadata:
            for rs in (ray_samples, *prop_ray_samples):
                self._mark_close_to_lidar(rs)
        return ray_samples, prop_ray_samples, prop_weights

    def _scale_pixel_area(self, ray_bundle: RayBundle) -> None:
        """camera rays cover rgb_upsample_factor^2 pixels, lidar rays one beam (models/neurad.py:702-709)"""
        up2 = float(self.config.rgb_upsample_factor**2)
        is_lidar = ray_bundle.metadata.get("is_lidar")
        if is_lidar is None:
            ray_bundle.pixel_area = ray_bundle.pixel_area * up2
        else:
            ray_bundle.pixel_area = torch.where(is_lidar, ray_bundle.pixel_area, ray_bundle.pixel_area * up2)

    def _render_weights(self, outputs, ray_samples):  # models/neurad.py:711-724 (no cpu placeholder: GPU only)
        if self.config.field.use_sdf:
            return nerfacc.render_weight_from_alpha(outputs[FieldHeadNames.ALPHA].squeeze(-1))[0]
        fr = ray_samples.frustums
        return nerfacc.render_weight_from_density(t_starts=fr.starts.squeeze(-1), t_ends=fr.ends.squeeze(-1),
                                                  sigmas=outputs[FieldHeadNames.DENSITY].squeeze(-1))[0]

    def _get_appearance_embedding(self, ray_bundle, features):
        """per-ray lerp of the two temporally adjacent embeddings of the ray's sensor (models/neurad.py:423-441)"""
        sensor = ray_bundle.metadata.get("sensor_idxs")
        if sensor is None:
            assert not self.training, "sensor_idxs must be present in metadata during training"
            sensor = torch.zeros_like(features[..., :1], dtype=torch.long)
        table = self.appearance_embedding.weight
        if not self.config.use_temporal_appearance:
            return ag.EmbeddingLerpFn.apply(table, sensor.reshape(-1).long(), None, None)
        n = self._num_embeds_per_sensor
        slot = ray_bundle.times / self._duration * n
        lo = slot.floor().clamp(0, n - 1)
        hi = (lo + 1).clamp(0, n - 1)
        frac = slot - lo
        base = sensor * n
        # e_lo * (1 - frac) + e_hi * frac in one kernel; its backward sums the R gradient rows into the few embedding
        # rows through LDS instead of torch's sort-based embedding_dense_backward (twice)
        return ag.EmbeddingLerpFn.apply(table, (lo + base).reshape(-1).long(), (hi + base).reshape(-1).long(),
                                        frac.reshape(-1))

    def _mark_close_to_lidar(self, rs) -> None:
        """metadata["is_close_to_lidar"] per sample (models/neurad.py:677-700): a lidar sample is "close" when it lies
        within carving_epsilon of the measured return, or -- for beams without a return -- anywhere inside the sensor's
        range; camera samples never are."""
        close, _, _ = ops.lidar_carving(*self._carving_inputs(rs))  # one pass instead of ~7 elementwise ops per level
        rs.metadata["is_close_to_lidar"] = close[..., None]

    def _carving_inputs(self, rs):
        """(starts, ends [R,S] views, per-ray is_lidar, did_return | None, measured distance, epsilon, non-return range)"""
        md, fr = rs.metadata, rs.frustums

        def edges(t):
            t = t[..., 0]
            return t if t.stride(-1) == 1 else t.contiguous()

        starts, ends = edges(fr.starts), edges(fr.ends)
        if starts.stride(0) != ends.stride(0):
            starts, ends = starts.contiguous(), ends.contiguous()
        ray = lambda t: t[:, 0, 0]  # noqa: E731  (per-ray metadata is broadcast over the samples)
        return (starts, ends, ray(md["is_lidar"]), ray(md["did_return"]) if "did_return" in md else None,
                ray(md["directions_norm"]), self.config.carving_epsilon, self.config.non_return_lidar_distance)

    # ---- get_nff_outputs (models/neurad.py:368-421) ------------------------------------------------
    def get_nff_outputs(self, ray_bundle: RayBundle, calc_lidar_losses: bool = False) -> Dict[str, Tensor]:
        if self.fused_eval_possible():
            return self.fused_nff_outputs(ray_bundle)
        if self.fused_training_possible():
            return self._fused_train_nff_outputs(ray_bundle, calc_lidar_losses)
        self._scale_pixel_area(ray_bundle)
        ray_samples, proposal_ray_samples, proposal_weights = self._get_ray_samples(ray_bundle)
        outputs = self.field(ray_samples)
        weights = self._render_weights(outputs, ray_samples)
        accumulation = self.renderer_accumulation(weights=weights[..., None])
        # the transmittance left behind the last sample is sky: it goes onto the last sample's features
        weights = torch.cat((weights[..., :-1], weights[..., -1:] + 1 - accumulation), dim=-1).unsqueeze(-1)
        features = self.renderer_feat(features=outputs[FieldHeadNames.FEATURE], weights=weights)
        if self.config.appearance_dim > 0:
            features = torch.cat([features, self._get_appearance_embedding(ray_bundle, features)], dim=-1)
        weights, ray_samples = weights[..., :-1, :], ray_samples[..., :-1]  # the sky sample plays no further role
        nff = {"features": features, "depth": self.renderer_depth(weights=weights, ray_samples=ray_samples),
               "accumulation": accumulation}
        lidar_terms = self.training and calc_lidar_losses
        for i, (pw, prs) in enumerate(zip(proposal_weights, proposal_ray_samples)):
            nff[f"prop_depth_{i}"] = self.renderer_depth(pw, prs)
            if lidar_terms:  # carving: lidar weight away from the measured surface
                nff[f"prop_weights_loss_{i}"] = ag.CarvingLossFn.apply(pw.squeeze(-1), *self._carving_inputs(prs))
        if self.training:
            nff["weights_list"] = proposal_weights + [weights]
            nff["ray_samples_list"] = proposal_ray_samples + [ray_samples]
        if lidar_terms:  # models/neurad.py:410-419
            md = ray_samples.metadata
            sel = (md["is_lidar"] & ~md["is_close_to_lidar"]).squeeze(-1).nonzero(as_tuple=True)
            nff["non_nearby_weights"] = weights[sel]
            first_lidar_ray = ray_bundle.metadata["is_lidar"].int().argmax()
            nff["non_nearby_lidar_ray_indices"] = sel[0] - first_lidar_ray
        return nff

    def forward(self, ray_bundle: RayBundle, calc_lidar_losses: bool = False):
        return self.get_nff_outputs(ray_bundle, calc_lidar_losses)
