"""NeuRADHashEncoding: static-world hash grid + per-actor grids with ZipNeRF-style down-weighting
(mirror of nerfstudio/field_components/neurad_encoding.py:34-304).

Static path: contraction (H3), gaussian (H2), lookup (H1) and rescale (H4) are ONE HIP kernel (nrhip_encode_fwd)
consuming per-ray origin/direction and per-sample [start,end]; nothing of shape [R,S,3] is materialised.
Actor path (H5): nrhip_actor_prepare (pose interpolation + line cull per ray) then nrhip_actor_encode (in-box test,
actor-grid lookup, overwrite) -- torch-path semantics (one 3-D grid per actor, use_4d_hashgrid=False).  Eval renders
actor scenes through the fused kernels instead (fields/neurad_field.py:render, models/neurad.py).  Training: the hit rows
are recomputed differentiably (``_actor_rows_with_grad``: pair positions + multi-grid lookup kernels with hand-written
backward) so actor grids and trajectories get the reference's gradients; rows overwritten by actors give the static
table no gradient."""
from __future__ import annotations

import weakref

from dataclasses import dataclass, field
from typing import Dict, Optional, Tuple

import torch
from torch import Tensor, nn

from .. import _lib
from .. import autograd as ag
from .. import ops
from .encodings import HashEncoding


@dataclass
class StaticSettings:  # neurad_encoding.py:34-45
    hashgrid_dim: int = 4
    num_levels: int = 8
    base_res: int = 32
    max_res: int = 8192
    log2_hashmap_size: int = 22


@dataclass
class ActorSettings:  # neurad_encoding.py:48-66
    flip_prob: float = 0.5
    actor_scale: float = 10.0
    hashgrid_dim: int = 4
    num_levels: int = 4
    base_res: int = 64
    max_res: int = 1024
    log2_hashmap_size: int = 17
    use_4d_hashgrid: bool = False  # the 4-D grid exists only in tiny-cuda-nn (SURVEY §8b caveat)


@dataclass
class NeuRADHashEncodingConfig:  # neurad_encoding.py:69-82
    static: StaticSettings = field(default_factory=StaticSettings)
    actor: ActorSettings = field(default_factory=ActorSettings)
    disable_actors: bool = False
    require_actor_grad: bool = True

    def setup(self, **kwargs):
        return NeuRADHashEncoding(self, **kwargs)


class _MaskRowsFn(torch.autograd.Function):
    """identity forward on the (already overwritten) feature rows; backward zeroes the gradient of the rows that
    were replaced by actor features, i.e. the autograd of ``features[ray_idx, sample_idx] = ...`` w.r.t. the
    static features (neurad_encoding.py:184-185)."""

    @staticmethod
    def forward(ctx, feats, hit):
        ctx.save_for_backward(hit)
        return feats

    @staticmethod
    def backward(ctx, g):
        (hit,) = ctx.saved_tensors
        return g.masked_fill(hit[:, None], 0.0), None


class _SpliceActorRowsFn(torch.autograd.Function):
    """``features[ray_idx, sample_idx] = actor_rows`` with the reference's autograd (neurad_encoding.py:184-185) and no
    sort / nonzero / host sync.  pairs: ``idx`` [P] flat sample index, ``winner`` [P] (exactly one per hit sample).
    Value: the winner's row replaces the static row.  Gradient: the replaced static rows get none; EVERY pair -- the
    shadowed actors of overlapping boxes too -- receives the gradient of its sample's row (index_put's backward hands it
    to all duplicates)."""

    @staticmethod
    def forward(ctx, feats, rows, idx, winner):
        n = feats.shape[0]
        buf = torch.empty((n + 1, feats.shape[1]), dtype=feats.dtype, device=feats.device)
        buf[:n] = feats
        buf.index_copy_(0, torch.where(winner, idx, n), rows)  # the shadowed pairs land in the spare row n
        ctx.save_for_backward(idx)
        return buf[:n]

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        g_rows = g.index_select(0, idx) if ctx.needs_input_grad[1] else None
        g_feats = g.index_fill(0, idx, 0.0) if ctx.needs_input_grad[0] else None
        return g_feats, g_rows, None, None


_STACKED_TABLES: "weakref.WeakKeyDictionary" = weakref.WeakKeyDictionary()  # encoding -> (key, stacked tables, bundle)


class NeuRADHashEncoding(nn.Module):
    def __init__(self, config: NeuRADHashEncodingConfig, dynamic_actors=None, static_scale: float = 1.0,
                 implementation: str = "hip") -> None:
        super().__init__()
        self.config, self.implementation = config, implementation
        self.actors = dynamic_actors
        self.static_scale = float(static_scale)  # the reference passes scene_box.aabb.max() (a 0-d tensor)
        s, a = config.static, config.actor
        self.static_grid = HashEncoding(num_levels=s.num_levels, min_res=s.base_res, max_res=s.max_res,
                                        log2_hashmap_size=s.log2_hashmap_size, features_per_level=s.hashgrid_dim,
                                        implementation=implementation)
        n_actors = 0 if dynamic_actors is None else int(getattr(dynamic_actors, "n_actors", 0))
        if n_actors and a.use_4d_hashgrid:
            # the reference's torch branch does exactly this (neurad_encoding.py:110-111: "4D hashgrid is not supported
            # with torch implementation, falling back multiple grids"); the 4-D grid exists only inside tiny-cuda-nn
            import warnings

            warnings.warn("use_4d_hashgrid=True is a tiny-cuda-nn feature; using one 3-D grid per actor like the "
                          "reference's torch implementation", stacklevel=2)
        self.actor_grids = nn.ModuleList([
            HashEncoding(num_levels=a.num_levels, min_res=a.base_res, max_res=a.max_res,
                         log2_hashmap_size=a.log2_hashmap_size, features_per_level=a.hashgrid_dim,
                         implementation=implementation) for _ in range(n_actors)])
        self.scene_repr_dim = self.static_grid.get_out_dim()
        # set by owners whose encoding is looked into several times per step (the proposal field: once per sampler round)
        self.share_actor_table_grads = False
        if n_actors and a.num_levels * a.hashgrid_dim > self.scene_repr_dim:
            raise ValueError("actor feature dim exceeds the static feature dim (F.pad would be negative)")

    def get_out_dim(self) -> int:
        return self.scene_repr_dim

    def get_param_groups(self, param_groups: Dict):
        param_groups["hashgrids"] += list(self.static_grid.parameters()) + list(self.actor_grids.parameters())

    def has_actors(self) -> bool:
        return (not self.config.disable_actors) and self.actors is not None and int(self.actors.n_actors) > 0

    def actor_spec(self) -> ops.ActorSpec:
        act = self.actors
        key = (act.actor_to_id._version, act.actor_to_id.data_ptr())
        if getattr(self, "_actor_ids", (None, None))[0] != key:  # buffer -> host list once, not once per forward
            self._actor_ids = (key, act.actor_to_id.tolist())
        ids = self._actor_ids[1]
        g0 = self.actor_grids[0]
        # The spec only holds POINTERS into the parameters: it stays valid across optimizer steps (in-place updates) and
        # is rebuilt when a tensor is re-assigned or when the box sizes / padding (-> bounds, a derived tensor) change.
        sizes, pad = act.actor_sizes, act.actor_padding
        pad_key = (pad._version, pad.data_ptr()) if isinstance(pad, Tensor) else pad
        skey = (act.actor_positions.data_ptr(), act.actor_rotations_6d.data_ptr(), act.unique_timestamps.data_ptr(),
                act.actor_present_at_time.data_ptr(), sizes._version, sizes.data_ptr(), pad_key,
                self.config.actor.actor_scale, tuple(self.actor_grids[i].hash_table.data_ptr() for i in ids))
        if getattr(self, "_actor_spec", (None, None))[0] != skey:
            self._actor_spec = (skey, ops.ActorSpec(
                act.unique_timestamps.float(), act.actor_positions.detach(), act.actor_rotations_6d.detach(),
                act.actor_present_at_time, act.actor_bounds().detach(), g0.spec,
                [self.actor_grids[i].hash_table.detach() for i in ids], self.config.actor.actor_scale))
        return self._actor_spec[1]

    def actor_edit(self) -> Optional[dict]:
        """DynamicActors.actor_editing when it changes anything: only outside training (get_boxes2world,
        dynamic_actors.py:261-265) and only if lateral, longitudinal or rotation is set (edit_boxes2world, :182-187)."""
        ed = getattr(self.actors, "actor_editing", None)
        if not ed or self.actors.training:
            return None
        if abs(ed.get("longitudinal", 0.0)) == 0.0 and abs(ed.get("lateral", 0.0)) == 0.0 and abs(ed.get("rotation", 0.0)) == 0.0:
            return None
        return ed

    def prepare_actors(self, origins, directions, pixel_area, starts, ends, times):
        """per-ray candidate lists (shared by every field evaluated on the same ray bundle)."""
        spec = self.actor_spec()
        return spec, ops.actor_prepare(spec, origins, directions, pixel_area, starts, ends, times, edit=self.actor_edit())

    def sample_ray_flip(self, origins) -> Optional[Tensor]:
        """-1 with prob flip_prob else +1, per ray, training only (neurad_encoding.py:212-215)."""
        p = self.config.actor.flip_prob
        if not (self.training and p > 1e-7):
            return None
        return torch.bernoulli(torch.full((origins.shape[0],), p, device=origins.device)) * -2 + 1

    def forward_rays(self, origins, directions, pixel_area, starts, ends, times: Optional[Tensor] = None):
        """-> (features [R*S, L*F], per-sample directions [R*S,3] or None when there are no actors)."""
        g = self.static_grid
        feats = ag.EncodeFn.apply(g.hash_table, g.spec, self.static_scale, origins, directions, pixel_area, starts, ends)
        if not self.has_actors():
            return feats, None
        if times is None:
            raise ValueError("dynamic actors need ray times")
        flip = self.sample_ray_flip(origins)
        with torch.no_grad():
            spec, cand = self.prepare_actors(origins, directions, pixel_area, starts, ends, times)
            merged = feats.detach().clone()
            dirs, hit = ops.actor_encode(spec, cand, origins, directions, pixel_area, starts, ends, merged, flip)
        if self.wants_actor_grad():
            with torch.no_grad():
                hits = ops.actor_hits(spec, cand, origins, directions, pixel_area, starts, ends)
            feats = self._actor_rows_with_grad(feats, hit, hits, origins, directions, pixel_area, starts, ends, times,
                                               flip)
        elif feats.requires_grad:
            # value = merged rows; gradient flows to the static features of the non-overwritten rows only
            feats = _MaskRowsFn.apply(feats + (merged - feats.detach()), hit >= 0)
        else:
            feats = merged
        return feats, dirs

    def wants_actor_grad(self) -> bool:
        """Does the actor branch have to be differentiable?  ``require_actor_grad`` governs the POSE gradient only
        (the reference wraps just ``_split_static_vs_actors`` in no_grad, neurad_encoding.py:174-176): the actor grids
        (and whatever consumes their features) train in both the main field and the proposal fields."""
        if not torch.is_grad_enabled():
            return False
        return any(gr.hash_table.requires_grad for gr in self.actor_grids) or (
            self.config.require_actor_grad and self.actors.actor_positions.requires_grad)

    def actor_pair_rows(self, hit, hits, origins, directions, pixel_area, starts, ends, times, flip):
        """Training path of the actor rows (B1): the kernels found WHICH samples lie in WHICH actor; the few hit rows
        are recomputed differentiably -- box-frame position through nrhip_actor_pair_positions_fwd/bwd (gradient to the
        trajectories when ``require_actor_grad``), the actor grids through MultiHashGridFn (table scatter-add, and
        nrhip_hashgrid_bwd_input for dL/dx).  -> None, or (idx [P] flat sample index, winner [P] bool: the actor the
        forward kernels used for that sample (highest index), rows [P, La*Fa] rescaled actor features)."""
        if self.actor_edit() is not None:
            # the differentiable rows are recomputed from the trajectories themselves (nrhip_actor_pair_positions_*), which
            # know nothing of the edit; the reference edits for rendering only (pipelines/ad_pipeline.py:476-480)
            raise NotImplementedError("eval-time actor edits (DynamicActors.actor_editing) with gradients enabled: "
                                      "render edited actors under torch.no_grad()")
        idx, act = ops.actor_pairs(hits)              # every (sample, containing actor), in (sample, slot) order
        if idx.shape[0] == 0:
            return None
        winner = act == hit[idx]                      # the row the forward actually used (highest actor index)
        # box-frame position + contraction of every pair in one kernel; its backward hands the trajectory parameters
        # their gradient (require_actor_grad governs exactly that: neurad_encoding.py:174-176)
        spec = self.actor_spec()
        pose_grad = torch.is_grad_enabled() and self.config.require_actor_grad
        with torch.set_grad_enabled(pose_grad):
            x01, cstd = ag.ActorPairPositionsFn.apply(self.actors.actor_positions, self.actors.actor_rotations_6d, spec,
                                                      origins, directions, pixel_area.reshape(-1), starts, ends, times, idx,
                                                      act, flip)
        act = act.long()
        ids = self.actors.actor_to_id[act]
        # _get_actor_features_slow loops over the actor ids; all actor grids share one shape, so one multi-grid
        # lookup (row i -> actor_grids[ids[i]]) does the same without the per-id launches and host syncs
        grid = self.actor_grids[0]
        tables = [g.hash_table for g in self.actor_grids]
        if self.share_actor_table_grads and len(tables) > 1 and torch.is_grad_enabled() and all(t.requires_grad for t in tables):
            stacked, bundle = self._stacked_actor_tables(tables)
            f = ag.MultiHashGridStackedFn.apply(x01, ids, grid.spec, stacked, bundle)
        else:
            f = ag.MultiHashGridFn.apply(x01, ids, grid.spec, *tables)
        w = 1 / (grid.scalings[None, :] * 2 * cstd[:, None]).clamp_min(1.0)
        return idx, winner, (f.view(-1, grid.num_levels, grid.features_per_level) * w[..., None]).flatten(1)

    def _stacked_actor_tables(self, tables):
        """the actor tables as one autograd-tracked tensor [A, rows, F] (ag.StackTablesFn), shared by every lookup of this
        step: a proposal field is evaluated once per sampler round, and with one node per round autograd would add the
        rounds' gradients on each of the A parameters (32 adds per c4 step); through the shared stack they meet in one add.
        Rebuilt once its backward has run, and whenever the tables changed (in place, or under autograd's feet: the
        optimizer kernels write through raw pointers and bump ops.TABLE_EPOCH)."""
        key = (ops.TABLE_EPOCH[0], tuple((t.data_ptr(), t._version) for t in tables))
        hit = _STACKED_TABLES.get(self)  # (not an attribute: a tensor with a grad_fn on the module would break deepcopy)
        if hit is None or hit[0] != key or hit[2].spent:
            bundle = ag.TableBundle(len(tables))
            hit = _STACKED_TABLES[self] = (key, ag.StackTablesFn.apply(bundle, *tables), bundle)
        return hit[1], hit[2]

    def _actor_rows_with_grad(self, feats, hit, hits, origins, directions, pixel_area, starts, ends, times, flip):
        """spliced in with index_put, whose autograd zeroes the static-table gradient of the replaced rows."""
        pr = self.actor_pair_rows(hit, hits, origins, directions, pixel_area, starts, ends, times, flip)
        if pr is None:
            return feats
        idx, winner, f = pr
        rows = torch.nn.functional.pad(f, (0, self.scene_repr_dim - f.shape[1]))
        return _SpliceActorRowsFn.apply(feats, rows, idx, winner)

    def forward(self, ray_samples, times=None, directions: Optional[Tensor] = None) -> Tuple[Tensor, Optional[Tensor]]:
        """Reference signature is forward(GaussiansStd, times, directions); here the frustums are passed directly
        (the gaussian is computed inside the kernel).  Returns (features [N, L*F], directions)."""
        from ..cameras.rays import per_ray

        fr = ray_samples.frustums
        o, d, a = per_ray(fr)
        t = times if times is not None else ray_samples.times
        t = None if t is None else (t[:, 0] if t.dim() == 3 else t).reshape(-1)
        feats, dirs = self.forward_rays(o, d, a, fr.starts[..., 0], fr.ends[..., 0], t)
        if dirs is not None:
            return feats, dirs.view(*fr.starts.shape[:-1], 3)
        return feats, directions
