"""NeuRADField / NeuRADProposalField on the HIP path (mirror of nerfstudio/fields/neurad_field.py:43-216).

Same constructor arguments, ``forward(ray_samples) -> Dict[FieldHeadNames, Tensor]``, ``get_density``,
``get_param_groups`` and state_dict names (``hashgrid.static_grid.hash_table``, ``mlp_geo.layers.k.*``,
``mlp_feature.layers.k.*``, ``sdf_to_density.beta``, ``density_decoder.weight``).

Two execution paths, both pure HIP:
  * no-grad (eval / render): ONE fused kernel -- nrhip_field_fwd per-sample, or nrhip_render_fwd when the
    caller wants composited rays (``render``; scenes with dynamic actors: nrhip_render_fwd_actors, per-sample
    table select in the same kernel);
  * grad-enabled (training), static scene: the same fused field kernel storing its activations, with the
    hand-written backward chained behind it in one autograd node (autograd.FieldTrainFn: feature-MLP and
    geometry-MLP data + weight gradients on the matrix cores, table gradient without memory-side atomics);
    the head (learnable-beta sigmoid / trunc_exp) stays a torch module;
  * grad-enabled with actors, or ``fused_training = False``: the reference's orchestration over operator-level
    autograd functions encode -> MLP -> SH -> MLP -> head.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, Optional, Tuple

import torch
from torch import Tensor, nn

from .. import autograd as ag
from .. import ops
from ..cameras.rays import RaySamples, per_ray, sample_times
from ..field_components.encodings import SHEncoding
from ..field_components.field_heads import FieldHeadNames
from ..field_components.mlp import MLP
from ..field_components.neurad_encoding import (ActorSettings, NeuRADHashEncoding, NeuRADHashEncodingConfig,
                                                StaticSettings)


class _TruncExp(torch.autograd.Function):  # field_components/activations.py:28-41
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return torch.exp(x)

    @staticmethod
    def backward(ctx, g):
        return g * torch.exp(ctx.saved_tensors[0].clamp(-15, 15))


trunc_exp = _TruncExp.apply


class SigmoidDensity(nn.Module):  # model_components/utils.py:21-41
    def __init__(self, init_val, beta_min=0.0001, learnable_beta=False):
        super().__init__()
        self.register_buffer("beta_min", torch.tensor(beta_min))
        self.register_parameter("beta", nn.Parameter(init_val * torch.ones(1), requires_grad=learnable_beta))
        # the buffer's value on the host: read once here (and after a checkpoint load), never per training step -- a
        # float(buffer) inside the step is a device->host sync between the queued sampler rounds and the field node
        self.beta_min_value = float(beta_min)

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)
        if prefix + "beta_min" in state_dict:
            self.beta_min_value = float(state_dict[prefix + "beta_min"])

    def get_beta(self):
        return self.beta.abs() + self.beta_min

    def forward(self, sdf: Tensor, beta=None) -> Tensor:
        return torch.sigmoid(-sdf * (self.get_beta() if beta is None else beta))


def get_normalized_directions(directions: Tensor) -> Tensor:  # fields/base_field.py:136-142
    return (directions + 1.0) / 2.0


@dataclass
class NeuRADFieldConfig:  # neurad_field.py:43-75
    grid: NeuRADHashEncodingConfig = field(
        default_factory=lambda: NeuRADHashEncodingConfig(require_actor_grad=True, actor=ActorSettings(flip_prob=0.25)))
    geo_hidden_dim: int = 32
    geo_num_layers: int = 2
    nff_hidden_dim: int = 32
    nff_num_layers: int = 3
    nff_out_dim: int = 32
    num_multisamples: int = 1
    use_sdf: bool = True
    sdf_beta: float = 20.0
    learnable_beta: bool = True

    def setup(self, **kwargs):
        return NeuRADField(self, **kwargs)


class NeuRADField(nn.Module):
    def __init__(self, config: NeuRADFieldConfig, actors=None, static_scale: float = 1.0,
                 implementation: str = "hip") -> None:
        super().__init__()
        if config.num_multisamples < 1:
            raise ValueError("num_multisamples >= 1")
        if config.num_multisamples != 1 and actors is not None and int(getattr(actors, "n_actors", 0)) > 0 \
                and not config.grid.disable_actors:
            # the reference raises in this configuration too, in the first batch with a sample inside a box: [P, M, 3] probe
            # positions meet one box transform per pair in a torch.bmm (neurad_encoding.py:197-198, cameras/lidars.py:559;
            # reproduced by oracle/check_multisample_actors_reference.py) -- there is no behaviour to match
            raise NotImplementedError("num_multisamples != 1 with dynamic actors: the reference fails on it as well "
                                      "(neurad_encoding.py:197-198 -> cameras/lidars.py:559 bmm of P*M points with P "
                                      "transforms; no NeuRAD config sets it, neurad_field.py:67)")
        self.config, self.implementation = config, implementation
        self.order_rays = False
        """Training forward: walk the batch in the cache-coherent order of ops.ray_order (computed per call).  Pays for
        incoherent batches (random / lidar rays: -10 % on the field forward), costs ~10 us for batches that are coherent
        as they come (camera patches)."""
        self.fused_training = True
        """Training forward through the fused field kernel + hand-chained backward (autograd.FieldTrainFn); False
        runs the reference orchestration over operator-level autograd functions (same numbers, 2x the launches)."""
        # built directly (not through config.grid.setup): ``config`` may be the REFERENCE's NeuRADFieldConfig, whose
        # grid._target is the reference's own NeuRADHashEncoding (field names are identical, neurad_encoding.py:34-82)
        self.hashgrid = NeuRADHashEncoding(config.grid, dynamic_actors=actors, static_scale=static_scale,
                                           implementation=implementation)
        self.geo_feat_dim = config.nff_out_dim
        self.mlp_geo = MLP(in_dim=self.hashgrid.get_out_dim(), num_layers=config.geo_num_layers,
                           layer_width=config.geo_hidden_dim, out_dim=self.geo_feat_dim + 1)
        self.direction_encoding = SHEncoding(levels=4)
        self.mlp_feature = MLP(in_dim=16 + self.geo_feat_dim, num_layers=config.nff_num_layers,
                               layer_width=config.nff_hidden_dim, out_dim=config.nff_out_dim)
        if config.use_sdf:
            self.sdf_to_density = SigmoidDensity(config.sdf_beta, learnable_beta=config.learnable_beta)

    def get_param_groups(self, param_groups: Dict):
        self.hashgrid.get_param_groups(param_groups)
        param_groups["fields"] += list(self.mlp_geo.parameters()) + list(self.mlp_feature.parameters())
        if self.config.use_sdf:
            param_groups["fields"] += list(self.sdf_to_density.parameters())

    # ---- fused path -----------------------------------------------------------------------------
    def fused_supported(self, with_actors: bool = False) -> bool:
        """Can the fused kernels evaluate this field?  ``with_actors``: the composited eval kernel also covers scenes
        with dynamic actors (per-sample table select, nrhip_render_fwd_actors) when the actor grids share the static
        grid's features per level (the reference's defaults); the other fused kernels cover the static scene only."""
        c, g = self.config, self.hashgrid.static_grid
        if c.num_multisamples != 1:  # M probes per frustum: the operator-level path (forward) averages their encodings
            return False
        if self.hashgrid.has_actors():
            ag_ = self.hashgrid.actor_grids[0]
            if not (with_actors and ag_.features_per_level == g.features_per_level and ag_.num_levels <= g.num_levels
                    and all(a.hash_table.dtype == g.hash_table.dtype for a in self.hashgrid.actor_grids)
                    and (g.num_levels, c.geo_hidden_dim) in ((8, 32), (8, 64), (16, 64))):
                return False
        return (g.get_out_dim() == 32 and g.num_levels % 4 == 0 and c.geo_num_layers == 2 and c.nff_num_layers == 3
                and c.geo_hidden_dim == c.nff_hidden_dim and c.geo_hidden_dim in (32, 64) and c.nff_out_dim == 32)

    def train(self, mode: bool = True):
        """a mode switch drops the cached host copy of beta: writes through ``beta.data`` (EMA / weight averaging, some
        optimizers) do not bump the version counter the cache is keyed on, and they happen between training and eval"""
        self._beta_cache = (None, 0.0)
        # ... and the re-laid-out eval tables (NRHIP_EVAL_RELAYOUT=1): HashGridAdam updates fp32 tables through raw
        # pointers, which does not bump the version the cache is keyed on -- train -> eval must rebuild them
        ops.clear_eval_tables()
        return super().train(mode)

    def invalidate_caches(self) -> None:
        """call after editing parameters through ``.data`` while staying in one mode"""
        self._beta_cache = (None, 0.0)
        self.hashgrid._actor_spec = (None, None)
        ops.clear_eval_tables()

    def _beta_value(self) -> float:
        """|beta| + beta_min as a host float, read from the device only when the parameter changed (an optimizer step or
        a checkpoint load bumps its version counter; ``train()`` / ``eval()`` / ``invalidate_caches()`` drop it) -- not
        once per eval chunk."""
        if not self.config.use_sdf:
            return 0.0
        b = self.sdf_to_density.beta
        key = (b._version, b.data_ptr())
        if getattr(self, "_beta_cache", (None, 0.0))[0] != key:
            self._beta_cache = (key, float(self.sdf_to_density.get_beta().detach()))
        return self._beta_cache[1]

    def field_spec(self) -> ops.FieldSpec:
        g = self.hashgrid.static_grid
        beta = self._beta_value()
        return ops.FieldSpec(g.spec, g.hash_table.detach(), self.hashgrid.static_scale,
                             [l.weight.detach() for l in self.mlp_geo.layers], [l.bias.detach() for l in self.mlp_geo.layers],
                             [l.weight.detach() for l in self.mlp_feature.layers],
                             [l.bias.detach() for l in self.mlp_feature.layers], use_sdf=self.config.use_sdf, beta=beta)

    @torch.no_grad()
    def render(self, origins, directions, pixel_area, starts, ends, return_weights=False, early_stop_eps: float = 0.0,
               order: Optional[Tensor] = None, times: Optional[Tensor] = None, actor_cand=None):
        """F1+C1+C2 in one kernel: -> features [R,32], depth [R,1], accumulation [R,1] (, weights [R,S]).
        early_stop_eps / order: see ops.render_fwd (eval-time ray termination; processing order from ops.ray_order).
        times [R]: needed when the scene has dynamic actors (their poses at the ray's time); actor_cand: candidate lists
        already computed for these rays (they depend on the ray's line only), else they are computed here."""
        if not self.fused_supported(with_actors=True):
            raise NotImplementedError("fused render kernel: configuration not instantiated; use forward() + renderers")
        if self.hashgrid.has_actors():
            if actor_cand is not None:
                spec, cand = self.hashgrid.actor_spec(), actor_cand
            elif times is None:
                raise ValueError("dynamic actors need ray times")
            else:
                spec, cand = self.hashgrid.prepare_actors(origins, directions, pixel_area, starts, ends, times)
            return ops.render_fwd_actors(self.field_spec(), spec, cand, origins, directions, pixel_area, starts, ends,
                                         return_weights, early_stop_eps=early_stop_eps, order=order)
        return ops.render_fwd(self.field_spec(), origins, directions, pixel_area, starts, ends, return_weights,
                              early_stop_eps=early_stop_eps, order=order)

    def render_train(self, origins, directions, pixel_area, edges, appearance=None, times: Optional[Tensor] = None,
                     actor_cand=None):
        """Training counterpart of ``render``: field -> learnable-beta SDF head -> weights -> compositing as ONE autograd
        node (autograd.NffRenderTrainFn) from the bin edges [R,S+1] (last edge = sky distance).  appearance: None or
        (embedding weight [E,A], sensor_idx [R,1] | None, times [R,1] | None, (duration, n_per_sensor, temporal)): the
        appearance embedding is written beside the features.  times [R]: needed with dynamic actors.
        -> features [R, 32 + A], depth [R,1], accumulation [R,1], weights of the non-sky samples [R,S-1]

        Dynamic actors (neurad_encoding.py:150-187): the kernels find which samples lie in which box; the differentiable
        actor branch (``actor_pair_rows``: box-frame positions with their pose gradient, one multi-grid lookup) produces the
        rows of those few samples, and the fused forward takes them -- and their box-frame view directions -- in place of
        the static lookup (nrhip_field_fwd_train_ovr).  The backward hands every (sample, actor) pair its sample's row
        gradient, as the reference's index_put does, and the static table nothing for those samples."""
        hg = self.hashgrid
        if not (self.fused_supported(with_actors=True) and self._fused_train_ok()):
            raise NotImplementedError("render_train: a fused-kernel configuration only (L * F = 32, 32- or 64-wide MLPs)")
        g = hg.static_grid
        emb, sensor, etimes, emb_cfg = appearance if appearance is not None else (None, None, None, (1.0, 1, False))
        order = ops.ray_order(origins, directions, hg.static_scale) if self.order_rays else None
        ovr = (None, None, None, None)
        if hg.has_actors():
            if times is None:
                raise ValueError("dynamic actors need ray times")
            ovr = self._actor_overrides(origins, directions, pixel_area.reshape(-1), edges, times.reshape(-1), actor_cand)
        # use_sdf = False (fields/neurad_field.py:149-151): the density head -- beta = None selects it in the node
        beta, beta_min = (self.sdf_to_density.beta, self.sdf_to_density.beta_min_value) if self.config.use_sdf else (None, 0.0)
        return ag.NffRenderTrainFn.apply(
            g.hash_table, g.spec, hg.static_scale, beta, beta_min, origins, directions,
            pixel_area.reshape(-1), edges, emb, sensor, etimes, emb_cfg, order, *ovr,
            *[t for l in self.mlp_geo.layers for t in (l.weight, l.bias)],
            *[t for l in self.mlp_feature.layers for t in (l.weight, l.bias)])

    def _actor_overrides(self, origins, directions, pixel_area, edges, times, actor_cand=None):
        """-> (ovr_row int32 [N]: row of the sample's WINNING actor (highest index containing it) or -1, rows [P,32],
        box-frame view directions [P,3], pair_idx [P]: flat sample index of every (sample, actor) pair) or four Nones"""
        hg = self.hashgrid
        starts, ends = edges[:, :-1], edges[:, 1:]
        N = starts.shape[0] * starts.shape[1]
        flip = hg.sample_ray_flip(origins)
        with torch.no_grad():
            # candidate lists depend on the ray's line only: one list per ray batch serves every field (as in eval)
            spec, cand = (hg.actor_spec(), actor_cand) if actor_cand is not None else hg.prepare_actors(
                origins, directions, pixel_area, starts, ends, times)
            scratch = torch.empty((N, hg.get_out_dim()), device=origins.device, dtype=torch.float32)  # (hit rows land here)
            dirs, hit = ops.actor_encode(spec, cand, origins, directions, pixel_area, starts, ends, scratch, flip)
            hits = ops.actor_hits(spec, cand, origins, directions, pixel_area, starts, ends)
        pr = hg.actor_pair_rows(hit, hits, origins, directions, pixel_area, starts, ends, times, flip)
        if pr is None:
            return None, None, None, None
        idx, winner, f = pr
        rows = torch.nn.functional.pad(f, (0, hg.get_out_dim() - f.shape[1])).contiguous()
        with torch.no_grad():
            P = idx.shape[0]
            slot = torch.where(winner, torch.arange(P, device=idx.device), -1)  # one winner per sample: amax picks it
            ov = torch.full((N,), -1, device=idx.device, dtype=torch.int64).scatter_reduce_(0, idx, slot, reduce="amax")
            pdirs = dirs.index_select(0, idx).contiguous()
        return ov.to(torch.int32), rows, pdirs, idx

    # ---- Field.forward (neurad_field.py:128-152) ------------------------------------------------
    def forward(self, ray_samples: RaySamples, compute_normals: bool = False) -> Dict[FieldHeadNames, Tensor]:
        fr = ray_samples.frustums  # this package's RaySamples or the reference's (cameras/rays.py:142-187)
        o, d, a = per_ray(fr)
        starts, ends = fr.starts[..., 0], fr.ends[..., 0]
        R, S = starts.shape
        needs_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        if not needs_grad and self.fused_supported():
            feature, sdf, head = ops.field_fwd(self.field_spec(), o, d, a, starts, ends)
            out = {FieldHeadNames.FEATURE: feature}
            if self.config.use_sdf:
                out[FieldHeadNames.SDF], out[FieldHeadNames.ALPHA] = sdf[..., None], head[..., None]
            else:
                out[FieldHeadNames.DENSITY] = head[..., None]
            return out
        if self.fused_training and self.fused_supported() and self._fused_train_ok():
            g = self.hashgrid.static_grid
            # beta only shapes the kernel's own alpha output, which training does not use (the torch head below owns
            # the learnable beta) -- reading it here would be a device->host sync per step
            feature, geo_out = ag.FieldTrainFn.apply(
                g.hash_table, g.spec, self.hashgrid.static_scale, self.config.use_sdf, 1.0, o, d, a, starts, ends,
                *[t for l in self.mlp_geo.layers for t in (l.weight, l.bias)],
                *[t for l in self.mlp_feature.layers for t in (l.weight, l.bias)],
                *([ops.ray_order(o, d, self.hashgrid.static_scale)] if self.order_rays else []))
            return self._heads(feature.view(R, S, self.config.nff_out_dim), geo_out.view(R, S, 1))
        M = self.config.num_multisamples
        if M == 1:
            features, sample_dirs = self.hashgrid.forward_rays(o, d, a, starts, ends, sample_times(ray_samples))
        else:
            # get_fast_isotropic_gaussian(M) (cameras/rays.py:109-124): probes at t_k = start + k step, step = (end - start) /
            # (M + 1), each with std_k = (area t_k^2 step)^(1/3) -- exactly the M = 1 gaussian of the interval (t_k - step,
            # t_k + step), which is what the encode kernel computes from an interval; the rescaled features are averaged
            # over k (neurad_encoding.py:297-304, mean(dim=-3)).  M launches of the M = 1 kernels, autograd through each.
            step = (ends - starts) / (M + 1)
            features, sample_dirs = None, None
            for k in range(1, M + 1):
                tk = starts + k * step
                fk, _ = self.hashgrid.forward_rays(o, d, a, (tk - step).contiguous(), (tk + step).contiguous(),
                                                   sample_times(ray_samples))
                features = fk if features is None else features + fk
            features = features / M
        geo = self.mlp_geo(features)
        geo_out, geo_embedding = geo[:, :1], geo[:, 1:]
        if sample_dirs is None:
            sh = self.direction_encoding(get_normalized_directions(d))  # per ray; broadcast over the samples
            sh = sh[:, None, :].expand(R, S, 16).reshape(-1, 16)
        else:  # samples inside actors carry box-frame directions (neurad_encoding.py:203-208)
            # Known deviation: SH is evaluated without gradient (the reference's SHEncoding.pytorch_fwd is @no_grad too
            # for its OUTPUT, encodings.py:797, so no gradient reaches actor_rotations_6d through the view direction
            # there either; only tcnn's SH would propagate it).
            sh = self.direction_encoding(get_normalized_directions(sample_dirs))
        feature = geo_embedding + self.mlp_feature(torch.cat([geo_embedding, sh], dim=-1))
        return self._heads(feature.view(R, S, self.config.nff_out_dim), geo_out.reshape(R, S, 1))

    def _heads(self, feature: Tensor, geo_out: Tensor) -> Dict[FieldHeadNames, Tensor]:
        out = {FieldHeadNames.FEATURE: feature}
        if self.config.use_sdf:
            out[FieldHeadNames.SDF] = geo_out
            out[FieldHeadNames.ALPHA] = self.sdf_to_density(geo_out)  # torch: beta may be learnable
        else:
            out[FieldHeadNames.DENSITY] = trunc_exp(geo_out)
        return out

    def _fused_train_ok(self) -> bool:
        """The fused training forward needs biases on every layer (fp32 or fp16-storage table: the kernel reads either,
        the table gradient is formed in fp32 and handed to autograd in the table's dtype)."""
        return all(l.bias is not None for l in list(self.mlp_geo.layers) + list(self.mlp_feature.layers))


@dataclass
class NeuRADProposalFieldConfig:  # neurad_field.py:155-179
    grid: NeuRADHashEncodingConfig = field(default_factory=lambda: NeuRADHashEncodingConfig(
        static=StaticSettings(log2_hashmap_size=20, num_levels=6, max_res=4096, base_res=128, hashgrid_dim=1),
        actor=ActorSettings(log2_hashmap_size=15, num_levels=4, base_res=64, max_res=1024, hashgrid_dim=1),
        require_actor_grad=False))
    hidden_dim: int = 16

    def setup(self, **kwargs):
        return NeuRADProposalField(self, **kwargs)


class NeuRADProposalField(nn.Module):
    def __init__(self, config: NeuRADProposalFieldConfig, actors=None, static_scale: float = 1.0,
                 implementation: str = "hip") -> None:
        super().__init__()
        self.config, self.implementation = config, implementation
        self.hashgrid = NeuRADHashEncoding(config.grid, dynamic_actors=actors, static_scale=static_scale,
                                           implementation=implementation)
        self.hashgrid.share_actor_table_grads = True  # evaluated once per sampler round: the rounds' gradients meet in one add
        self.density_decoder = nn.Linear(self.hashgrid.get_out_dim(), 1, bias=False)

    def get_param_groups(self, param_groups: Dict):
        self.hashgrid.get_param_groups(param_groups)
        param_groups["fields"] += list(self.density_decoder.parameters())

    def fused_sampler_supported(self) -> bool:
        """Can the fused proposal sampler evaluate this field?  Always for the static scene; with dynamic actors when their
        grids have one feature per level, fp32 tables and at most the static grid's levels (the reference's defaults).  The
        static table may be fp32 or fp16 storage."""
        hg = self.hashgrid
        if not hg.has_actors():
            return True
        g, a = hg.static_grid, hg.actor_grids[0]
        return (a.features_per_level == 1 and a.num_levels <= g.num_levels
                and all(t.hash_table.dtype == torch.float32 for t in hg.actor_grids))

    def proposal_spec(self) -> ops.ProposalSpec:
        g = self.hashgrid.static_grid
        return ops.ProposalSpec(g.spec, g.hash_table.detach(), self.hashgrid.static_scale,
                                self.density_decoder.weight.detach())

    def get_density(self, ray_samples: RaySamples, actor_cand=None) -> Tuple[Tensor, None]:
        """neurad_field.py:208-213, one kernel: gaussian -> contraction -> 6-level lookup -> rescale -> dot -> exp.
        actor_cand: candidate lists already computed for these rays (they depend on the ray's line only)."""
        fr = ray_samples.frustums
        o, d, a = per_ray(fr)
        hg = self.hashgrid
        g = hg.static_grid
        starts, ends = fr.starts[..., 0], fr.ends[..., 0]
        need_graph = torch.is_grad_enabled() and (g.hash_table.requires_grad or self.density_decoder.weight.requires_grad)
        if need_graph:
            dens = ag.ProposalDensityFn.apply(g.hash_table, self.density_decoder.weight, g.spec, hg.static_scale, o, d, a,
                                              starts, ends)
        else:  # eval / no_grad / frozen: nothing to save for a backward
            dens = ops.proposal_density_fwd(self.proposal_spec(), o, d, a, starts, ends)
        if hg.has_actors():
            times = sample_times(ray_samples)
            if times is None:
                raise ValueError("dynamic actors need ray times")
            flip = hg.sample_ray_flip(o)
            with torch.no_grad():  # geometry of the actor branch: never differentiated here (require_actor_grad=False)
                spec, cand = (hg.actor_spec(), actor_cand) if actor_cand is not None else hg.prepare_actors(
                    o, d, a, starts, ends, times)
                merged = dens.detach().clone()
                hit_actor = ops.actor_density(spec, cand, o, d, a, starts, ends, self.density_decoder.weight.detach(), merged,
                                              flip, return_actor=True)
            if hg.wants_actor_grad() or (need_graph and self.density_decoder.weight.requires_grad):
                dens = self._splice_actor_density(dens, hit_actor, spec, cand, o, d, a, starts, ends, times, flip)
            else:
                dens = torch.where(hit_actor >= 0, merged, dens)
        return dens[..., None], None

    def _splice_actor_density(self, dens, hit_actor, spec, cand, o, d, a, starts, ends, times, flip):
        """Differentiable actor rows of the proposal density.  ``require_actor_grad=False`` (neurad_field.py:177) only
        keeps the POSES out of the graph (neurad_encoding.py:174-176); the actor grids and the decoder are trained
        through the in-box samples exactly as through the static ones: density = trunc_exp(decoder(pad(actor_feat)))."""
        hg = self.hashgrid
        with torch.no_grad():
            hits = ops.actor_hits(spec, cand, o, d, a, starts, ends)
            # (hit_actor: the actor nrhip_actor_density used, the highest index containing the sample -- what
            # hits.max(-1) would find again with a 0.25 ms reduction over [N, K])
        pr = hg.actor_pair_rows(hit_actor.reshape(-1), hits, o, d, a, starts, ends, times, flip)
        if pr is None:
            return dens
        idx, winner, rows = pr
        w = self.density_decoder.weight[0, : rows.shape[1]]     # features are zero-padded up to the static width
        # One kernel each way (autograd.ActorDensitySpliceFn).  The winner of a sample writes trunc_exp(rows . w); overlapping
        # boxes: the reference's features[ray, sample] = ... hands the merged row's gradient to every duplicate index
        # (neurad_encoding.py:184-185) with the value unchanged -- only the shadowed actors' FEATURES see that gradient, the
        # decoder's own comes from the winner.  No boolean indexing, no host read.
        return ag.ActorDensitySpliceFn.apply(dens, rows, w, idx, winner)

    def get_outputs(self, ray_samples, density_embedding=None) -> dict:
        return {}
