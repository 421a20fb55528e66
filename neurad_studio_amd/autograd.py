"""torch.autograd.Function wrappers over the HIP ops (operator-level training path).

Every Function runs under ``custom_fwd(cast_inputs=float32)`` like the reference's ``trunc_exp``
(field_components/activations.py:31-41), because the trainer runs under autocast + GradScaler
(engine/trainer.py:550-553).  Gradients flow to hash tables, MLP weights/biases, the proposal decoder,
densities/alphas and features, to the actor trajectories, and -- when the ray bundle's origins / directions require grad
(a camera optimizer moved them, cameras/camera_optimizers.py:173-182) -- to the rays: `_ray_grads` below, one kernel per
hash grid the rays were encoded with (nrhip_encode_bwd_rays; view directions carry none: SHEncoding.pytorch_fwd is
@torch.no_grad in the parity target, encodings.py:797).
"""
from __future__ import annotations

import torch
from torch.amp import custom_bwd, custom_fwd

from . import ops


_E15, _EM15 = 3269017.3724721107, 3.0590232050182579e-07  # exp(+-15): trunc_exp's backward clamp (activations.py:37-41)


def _ray_grads(need_o: bool, need_d: bool, spec, table, scale, o, d, a, starts, ends, genc):
    """(dL/d origins, dL/d directions) of a static encoding from dL/d(rescaled features), None where not needed"""
    if not (need_o or need_d):
        return None, None
    go, gd = ops.encode_bwd_rays(spec, table, scale, o, d, a, starts, ends, genc)
    return (go if need_o else None), (gd if need_d else None)


def _proposal_genc(dens, gdens, decoder_weight):
    """dL/d(rescaled level features) [N, L] of density = trunc_exp(features . decoder) from dL/d density"""
    gl = gdens.reshape(-1) * dens.reshape(-1).clamp(_EM15, _E15)
    return gl[:, None] * decoder_weight.reshape(1, -1)


class HashGridFn(torch.autograd.Function):
    """HashEncoding.pytorch_fwd (encodings.py:425-466): x [N,3] in [0,1], table [L*T,F] -> [N, L*F]."""

    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, x, table, spec):
        ctx.spec, ctx.table_dtype = spec, table.dtype
        ctx.save_for_backward(x, table)
        return ops.hashgrid_fwd(spec, table, x)

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, g):
        x, table = ctx.saved_tensors
        g = g.contiguous()
        gt = ops.hashgrid_bwd(ctx.spec, None, x, g) if ctx.needs_input_grad[1] else None
        gx = ops.hashgrid_bwd_input(ctx.spec, table, x, g) if ctx.needs_input_grad[0] else None  # actor poses only
        return gx, _like_param(gt, ctx.table_dtype), None


class ActorPairPositionsFn(torch.autograd.Function):
    """Box-frame contracted positions of (sample, actor) pairs, differentiable w.r.t. the trajectory parameters
    (interpolate_trajectories_6d -> rotation_6d_to_matrix -> pose inverse -> transform -> flip -> contraction, see
    csrc/actors.hip).  args: actor_positions [Tn,A,3], actor_rotations_6d [Tn,A,6] (the tensors inside ``spec``, passed
    so that autograd sees them), spec, origins, directions, pixel_area, starts, ends, times, sample_idx, actor_idx, flip.
    -> x01 [P,3], cstd [P]"""

    @staticmethod
    def forward(ctx, positions, rotations_6d, spec, origins, directions, pixel_area, starts, ends, times, sample_idx,
                actor_idx, flip):
        ctx.spec, ctx.flip = spec, flip
        ctx.save_for_backward(origins, directions, pixel_area, starts, ends, times, sample_idx, actor_idx)
        return ops.actor_pair_positions(spec, origins, directions, pixel_area, starts, ends, times, sample_idx, actor_idx,
                                        flip)

    @staticmethod
    def backward(ctx, g_x01, g_cstd):
        o, d, a, s, e, times, si, ai = ctx.saved_tensors
        need_o, need_d = ctx.needs_input_grad[3], ctx.needs_input_grad[4]
        if need_o or need_d:  # the rays move with a camera optimizer: the in-box samples' share of its gradient
            gp, gr, go, gd = ops.actor_pair_positions_bwd(ctx.spec, o, d, a, s, e, times, si, ai, ctx.flip,
                                                          g_x01.contiguous(), g_cstd.contiguous(), ray_grads=True)
            return (gp, gr, None, go if need_o else None, gd if need_d else None) + (None,) * 7
        gp, gr = ops.actor_pair_positions_bwd(ctx.spec, o, d, a, s, e, times, si, ai, ctx.flip, g_x01.contiguous(),
                                              g_cstd.contiguous())
        return (gp, gr) + (None,) * 10


class MultiHashGridFn(torch.autograd.Function):
    """HashEncoding.pytorch_fwd over several grids of one shape: row i looks into tables[grid_id[i]] -- the per-actor
    grids of NeuRADHashEncoding in one launch (`_get_actor_features_slow` loops over actor ids,
    neurad_encoding.py:270-295).  args: x [N,3], grid_id [N], spec, *tables."""

    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, x, grid_id, spec, *tables):
        ctx.spec = spec
        grid_id = grid_id.to(torch.int32)
        # which grids the batch touches: read HERE (the caller has just read the batch size from the device, the queue is
        # empty) so that the backward, in the middle of a full queue, does not stall on a device->host read
        # (inside a custom Function's forward grad mode is always off: ctx.needs_input_grad is what tells a recording
        # call from an eval / no_grad one, where the tables' requires_grad stays True but no backward will come)
        ctx.present = ops.grids_present(grid_id, len(tables)) if any(ctx.needs_input_grad[3:]) else None
        ctx.save_for_backward(x, grid_id, *tables)
        return ops.hashgrid_multi_fwd(spec, tables, grid_id, x)

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, g):
        x, grid_id, *tables = ctx.saved_tensors
        g = g.contiguous()
        gts = ops.hashgrid_multi_bwd(ctx.spec, len(tables), grid_id, x, g, present=ctx.present, out_dtype=tables[0].dtype) \
            if any(ctx.needs_input_grad[3:]) else [None] * len(tables)
        gx = ops.hashgrid_multi_bwd_input(ctx.spec, tables, grid_id, x, g) if ctx.needs_input_grad[0] else None
        return (gx, None, None, *gts)


class TableBundle:
    """what the consumers of one ``StackTablesFn`` output share: which grids ANY of their batches touched (a grid none
    touched gets no gradient at all -- the reference's per-id loop never evaluates it, so torch.optim.Adam leaves its
    moments alone), and whether the node's backward has run (its graph is gone: build a new one)"""

    def __init__(self, n_grids: int) -> None:
        self.present = [False] * n_grids
        self.spent = False


class StackTablesFn(torch.autograd.Function):
    """tables of one shape -> ONE tensor [A, rows, F] (a copy), for a set of grids that several nodes of one step look into
    (the proposal field's actor grids: once per sampler round).  The consumers' gradients then meet on the stacked tensor --
    one add of the block -- instead of on each of the A parameters (autograd's AccumulateGrad: A adds per extra consumer; 32
    per c4 step).  Backward: table a gets row a of the block, or None when no consumer's batch touched it."""

    @staticmethod
    def forward(ctx, bundle, *tables):
        ctx.bundle = bundle
        return torch.stack(tables)

    @staticmethod
    def backward(ctx, g):
        ctx.bundle.spent = True
        return (None, *[g[a] if p else None for a, p in enumerate(ctx.bundle.present)])


class MultiHashGridStackedFn(torch.autograd.Function):
    """MultiHashGridFn over a stacked table set (StackTablesFn): the table gradient is ONE block [A, rows, F].
    args: x [N,3], grid_id [N], spec, stacked [A, rows, F], bundle."""

    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, x, grid_id, spec, stacked, bundle):
        ctx.spec = spec
        grid_id = grid_id.to(torch.int32)
        ctx.present = ops.grids_present(grid_id, stacked.shape[0]) if ctx.needs_input_grad[3] else None  # (see MultiHashGridFn)
        if ctx.present is not None:
            bundle.present = [bool(a) or bool(b) for a, b in zip(bundle.present, ctx.present)]
        ctx.save_for_backward(x, grid_id, stacked)
        return ops.hashgrid_multi_fwd(spec, list(stacked.unbind(0)), grid_id, x)

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, g):
        x, grid_id, stacked = ctx.saved_tensors
        g = g.contiguous()
        block = ops.hashgrid_multi_bwd(ctx.spec, stacked.shape[0], grid_id, x, g, present=ctx.present, out_dtype=stacked.dtype,
                                       dense_block=True) if ctx.needs_input_grad[3] else None
        gx = ops.hashgrid_multi_bwd_input(ctx.spec, list(stacked.unbind(0)), grid_id, x, g) if ctx.needs_input_grad[0] else None
        return gx, None, None, block, None


class ActorDensitySpliceFn(torch.autograd.Function):
    """Proposal density with the in-box samples' values spliced in (fields/neurad_field.py:208-213 over
    neurad_encoding.py:150-187): dens [N] static density, rows [P, La] rescaled actor features of the P (sample, actor)
    pairs, weight [La] the decoder's first La columns, idx [P] flat sample index, winner [P] bool.  -> density [N].
    One kernel each way (csrc/actors.hip actor_density_splice_*): winners through trunc_exp, shadowed pairs of overlapping
    boxes with the reference's duplicate-index gradient, decoder gradient from the winners."""

    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, dens, rows, weight, idx, winner):
        out = dens.detach().reshape(-1).clone()
        rows, weight = rows.detach().contiguous(), weight.detach().contiguous()
        logit = ops.actor_density_splice_fwd(out, rows, weight, idx, winner)
        ctx.save_for_backward(rows, weight, idx, winner, logit, out)
        ctx.shape = dens.shape
        return out.view(dens.shape)

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, g):
        rows, weight, idx, winner, logit, out = ctx.saved_tensors
        g_dens, g_rows, g_w = ops.actor_density_splice_bwd(rows, weight, idx, winner, logit, out, g.contiguous())
        return (g_dens.view(ctx.shape) if ctx.needs_input_grad[0] else None, g_rows if ctx.needs_input_grad[1] else None,
                g_w if ctx.needs_input_grad[2] else None, None, None)


class EncodeFn(torch.autograd.Function):
    """NeuRADHashEncoding static path, fused H2->H3->H1->H4 (neurad_encoding.py:164-169,265-268,297-304)."""

    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, table, spec, static_scale, origins, directions, pixel_area, starts, ends):
        ctx.spec, ctx.scale, ctx.table_dtype = spec, static_scale, table.dtype
        ctx.rays = ctx.needs_input_grad[3] or ctx.needs_input_grad[4]
        ctx.save_for_backward(origins, directions, pixel_area, starts, ends, *([table] if ctx.rays else []))
        return ops.encode_fwd(spec, table, static_scale, origins, directions, pixel_area, starts, ends)

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, g):
        o, d, a, s, e, *tab = ctx.saved_tensors
        g = g.contiguous()
        gt = ops.encode_bwd(ctx.spec, ctx.scale, o, d, a, s, e, g, out_dtype=ctx.table_dtype) if ctx.needs_input_grad[0] else None
        go, gd = (None, None) if not ctx.rays else _ray_grads(ctx.needs_input_grad[3], ctx.needs_input_grad[4], ctx.spec,
                                                              tab[0], ctx.scale, o, d, a, s, e, g)
        return _like_param(gt, ctx.table_dtype), None, None, go, gd, None, None, None


class FieldTrainFn(torch.autograd.Function):
    """NeuRADField.forward for the static scene in ONE kernel, with the hand-written backward chained behind it
    (neurad_field.py:128-152).  Forward = the fused field kernel storing its activations; backward = feature-MLP
    gradients -> residual -> geometry-MLP gradients -> table gradient.  Replaces EncodeFn + 2x MLPFn + SH + concat.

    args: table, spec, static_scale, use_sdf, beta, o, d, area, starts, ends, gw0, gb0, gw1, gb1, fw0, fb0, fw1, fb1, fw2, fb2
    returns feature [N,32], geo_out [N,1] (sdf, or the pre-exp density logit)."""

    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, table, spec, static_scale, use_sdf, beta, origins, directions, pixel_area, starts, ends, *params):
        params, order = params[:10], (params[10] if len(params) > 10 else None)  # optional processing order (ops.ray_order)
        gw, gb, fw, fb = list(params[0:4:2]), list(params[1:4:2]), list(params[4:10:2]), list(params[5:10:2])
        fs = ops.FieldSpec(spec, table, static_scale, gw, gb, fw, fb, use_sdf=use_sdf, beta=beta)
        (feature, geo_out, _head), (enc, hg, xf, hf) = ops.field_fwd_train(fs, origins, directions, pixel_area, starts, ends,
                                                                           order=order)
        ctx.has_order, ctx.table_dtype = order is not None, table.dtype
        ctx.spec, ctx.scale = spec, static_scale
        ctx.rays = ctx.needs_input_grad[5] or ctx.needs_input_grad[6]
        ctx.save_for_backward(origins, directions, pixel_area, starts, ends, enc, hg, xf, hf, *params,
                              *([table] if ctx.rays else []))
        return feature, geo_out[:, None]

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, g_feature, g_geo_out):
        o, d, a, s, e, enc, hg, xf, hf, *params = ctx.saved_tensors
        rays = None
        if ctx.rays:
            params, rays = params[:-1], (params[-1], ctx.needs_input_grad[5], ctx.needs_input_grad[6])
        gt, grads, _, (go, gd) = _field_backward(ctx.spec, ctx.scale, ctx.table_dtype, ctx.needs_input_grad[0], o, d, a, s, e,
                                                 enc, hg, xf, hf, params, g_feature.contiguous(), g_geo_out, rays=rays)
        return (gt, None, None, None, None, go, gd, None, None, None, *grads, *([None] if ctx.has_order else []))


def _field_backward(spec, scale, table_dtype, need_table, o, d, a, starts, ends, enc, hg, xf, hf, params, g_feature, g_geo_out,
                    override=None, rays=None):
    """Backward of the fused field forward from (dL/dfeature [N,32], dL/dgeo_out [N]): feature-MLP gradients -> residual ->
    geometry-MLP gradients -> table gradient.  -> (grad table or None, the ten MLP parameter gradients in argument order,
    gradient of the override rows or None, (dL/d origins, dL/d directions) or Nones).  override = (ovr_row [N], pair_idx
    [P]): samples whose encoding row came from the caller (actor boxes) hand dL/d enc to those rows -- every (sample, actor)
    pair gets its sample's row, as the reference's index_put does (neurad_encoding.py:184-185) -- and send nothing to the
    static table.  rays = (table, need_o, need_d): the bundle's rays require grad (camera optimizer) -> the static samples'
    dL/d enc also goes back to the ray through the positions (`_ray_grads`; the overridden rows are zero by then: their
    share comes through ActorPairPositionsFn)."""
    gw, gb, fw, fb = list(params[0:4:2]), list(params[1:4:2]), list(params[4:10:2]), list(params[5:10:2])
    # feature = embedding + mlp_feature([embedding | sh])
    if ops.field_feature_bwd_supported(fw, fb):
        # one kernel: weight gradients + the geometry MLP's output gradient (residual add and column 0 included)
        g_geo, gfw, gfb = ops.field_feature_bwd(xf, hf, g_feature, g_geo_out, fw, fb)
    else:
        gxf, gfw, gfb = ops.mlp_bwd(xf, hf, g_feature, fw, fb)
        g_geo = torch.empty((g_feature.shape[0], 33), device=g_feature.device, dtype=torch.float32)
        g_geo[:, 0] = g_geo_out.reshape(-1)
        torch.add(g_feature, gxf[:, :32], out=g_geo[:, 1:])  # residual: feature = embedding + mlp_feature(...)
    genc, ggw, ggb = ops.mlp_bwd(enc, hg, g_geo, gw, gb)
    g_rows = None
    if override is not None:
        ovr_row, pair_idx = override
        g_rows = genc.index_select(0, pair_idx)
        # exactly-zero rows send no records (encode_bwd_binned: prep).  The overridden samples are the pair list's samples
        # (every sample of a pair has a winning pair): P rows written, not a pass over all N (85 us at 2 M samples)
        genc.index_fill_(0, pair_idx, 0.0)
    gt = _like_param(ops.encode_bwd(spec, scale, o, d, a, starts, ends, genc, out_dtype=table_dtype), table_dtype) \
        if need_table else None
    grads = [ggw[0], ggb[0], ggw[1], ggb[1], gfw[0], gfb[0], gfw[1], gfb[1], gfw[2], gfb[2]]
    god = (None, None) if rays is None else _ray_grads(rays[1], rays[2], spec, rays[0], scale, o, d, a, starts, ends, genc)
    return gt, grads, g_rows, god


class MLPFn(torch.autograd.Function):
    """MLP.pytorch_fwd (mlp.py:159-178).  args: x, n_layers, w0, b0, w1, b1, ... (b may be None)."""

    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, x, n_layers, *params):
        ws, bs = list(params[0::2]), list(params[1::2])
        y, hidden = ops.mlp_fwd(x, ws, bs, save_hidden=True)
        ctx.n = n_layers
        ctx.has_bias = [b is not None for b in bs]
        ctx.save_for_backward(x, hidden if hidden is not None else x.new_empty(0), *ws, *[b for b in bs if b is not None])
        return y

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, gy):
        n = ctx.n
        x, hidden, *rest = ctx.saved_tensors
        ws = rest[:n]
        bit = iter(rest[n:])
        bs = [next(bit) if hb else None for hb in ctx.has_bias]
        gx, gws, gbs = ops.mlp_bwd(x, hidden if hidden.numel() else None, gy.contiguous(), ws, bs,
                                   need_grad_x=ctx.needs_input_grad[0])
        out = [gx, None]
        for k in range(n):
            out += [gws[k], gbs[k]]
        return tuple(out)


def _like_param(grad, dtype):
    """table gradients are accumulated in fp32 by the kernels; autograd wants the parameter's dtype (fp16-storage tables,
    BASELINE config 5: the optimizer keeps what precision it needs, see optim.HashGridAdam)"""
    return grad if grad is None or grad.dtype == dtype else grad.to(dtype)


def mlp(x, weights, biases):
    params = []
    for w, b in zip(weights, biases):
        params += [w, b]
    return MLPFn.apply(x, len(weights), *params)


class WeightFromAlphaFn(torch.autograd.Function):
    """nerfacc.render_weight_from_alpha, dense mode (call site models/neurad.py:716-717)."""

    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, alphas):
        ctx.save_for_backward(alphas)
        w, t = ops.render_weight_from_alpha(alphas)
        return w, t

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, gw, gt):
        (a,) = ctx.saved_tensors
        return ops.render_weight_from_alpha_bwd(a, gw.contiguous(), None if gt is None else gt.contiguous())


class WeightFromDensityFn(torch.autograd.Function):
    """nerfacc.render_weight_from_density, dense mode (models/neurad.py:718-723).  Gradient w.r.t. sigmas."""

    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, t_starts, t_ends, sigmas):
        ctx.save_for_backward(t_starts, t_ends, sigmas)
        w, t, a = ops.render_weight_from_density(t_starts, t_ends, sigmas)
        ctx.mark_non_differentiable(t, a)
        return w, t, a

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, gw, gt, ga):
        s, e, sg = ctx.saved_tensors
        return None, None, ops.render_weight_from_density_bwd(s, e, sg, gw.contiguous())


class WeightsFromDensityFn(torch.autograd.Function):
    """RaySamples.get_weights (cameras/rays.py:188-210)."""

    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, deltas, densities):
        ctx.save_for_backward(deltas, densities)
        return ops.weights_from_density(deltas, densities)

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, gw):
        d, s = ctx.saved_tensors
        return None, ops.weights_from_density_bwd(d, s, gw.contiguous())


class AccumulateFn(torch.autograd.Function):
    """nerfacc.accumulate_along_rays dense mode: sum_S w * v."""

    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, weights, values):
        ctx.save_for_backward(weights, values)
        return ops.accumulate_along_rays(weights, values)

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, g):
        w, v = ctx.saved_tensors
        if v is None:
            return g.expand_as(w), None
        return ops.accumulate_along_rays_bwd(w, v, g.contiguous(), ctx.needs_input_grad[0], ctx.needs_input_grad[1])


class CarvingLossFn(torch.autograd.Function):
    """sum((w * (is_lidar & ~is_close))^2) over a level (models/neurad.py:399-408) -- value and gradient from ONE pass that
    recomputes the mask from the sample edges.  args: weights [R,S], starts, ends [R,S], is_lidar, did_return, distance
    (per ray), carving_epsilon, non_return_lidar_distance."""

    @staticmethod
    def forward(ctx, weights, starts, ends, is_lidar, did_return, distance, eps, non_return):
        _, loss, gw = ops.lidar_carving(starts, ends, is_lidar, did_return, distance, eps, non_return,
                                        weights=weights.contiguous(), want_mask=False)
        ctx.save_for_backward(gw)
        return loss.sum()

    @staticmethod
    def backward(ctx, g):
        (gw,) = ctx.saved_tensors
        return (gw * g,) + (None,) * 7


class EmbeddingLerpFn(torch.autograd.Function):
    """C3 (models/neurad.py:423-441): per-ray lerp of two embedding rows; gradient to the embedding table only."""

    @staticmethod
    def forward(ctx, weight, idx_lo, idx_hi, frac):
        ctx.save_for_backward(idx_lo, idx_hi, frac)
        ctx.n_embed = weight.shape[0]
        return ops.embedding_lerp(weight, idx_lo, idx_hi, frac)

    @staticmethod
    def backward(ctx, g):
        lo, hi, fr = ctx.saved_tensors
        return ops.embedding_lerp_bwd(g.contiguous(), lo, hi, fr, ctx.n_embed), None, None, None


class CompositeFn(torch.autograd.Function):
    """get_nff_outputs compositing (models/neurad.py:377-395,727-734): -> features [R,C], depth [R,1], acc [R,1]."""

    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, weights, features, starts, ends):
        ctx.save_for_backward(weights, features, starts, ends)
        return ops.composite_fwd(weights, features, starts, ends)

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, gf, gd, ga):
        w, f, s, e = ctx.saved_tensors
        gw, gfe = ops.composite_bwd(w, f, s, e, gf.contiguous(), gd, ga, need_grad_features=ctx.needs_input_grad[1])
        return gw, gfe, None, None


class ProposalDensityFn(torch.autograd.Function):
    """NeuRADProposalField.get_density (fields/neurad_field.py:208-213), trunc_exp backward included."""

    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, table, decoder_weight, spec, static_scale, origins, directions, pixel_area, starts, ends):
        ps = ops.ProposalSpec(spec, table, static_scale, decoder_weight)
        if not any(ctx.needs_input_grad):  # eval / frozen, rays fixed: nothing to save
            return ops.proposal_density_fwd(ps, origins, directions, pixel_area, starts, ends)
        dens, lf = ops.proposal_density_fwd(ps, origins, directions, pixel_area, starts, ends, save_features=True)
        ctx.ps = ps
        ctx.save_for_backward(origins, directions, pixel_area, starts, ends, dens, lf)
        return dens

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, g):
        o, d, a, s, e, dens, lf = ctx.saved_tensors
        g = g.contiguous()
        gt = gdec = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:  # (frozen tables under a camera optimizer: rays only)
            gt, gdec = ops.proposal_density_bwd(ctx.ps, o, d, a, s, e, dens, g, level_features=lf)
            gt, gdec = _like_param(gt, ctx.ps.table.dtype), gdec.reshape(ctx.ps.decoder_weight.shape)
        go, gd = (None, None)
        if ctx.needs_input_grad[4] or ctx.needs_input_grad[5]:
            go, gd = _ray_grads(ctx.needs_input_grad[4], ctx.needs_input_grad[5], ctx.ps.grid, ctx.ps.table,
                                ctx.ps.static_scale, o, d, a, s, e, _proposal_genc(dens, g, ctx.ps.decoder_weight))
        return gt, gdec, None, None, go, gd, None, None, None


class InterlevelLossFn(torch.autograd.Function):
    """One proposal level of zipnerf_interlevel_loss (model_components/losses.py:672-705), mean over rays.
    Gradient to the proposal weights only: the fine histogram is detached in the reference (losses.py:678-679)."""

    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, c, w, cp, wp, pulse_width):
        loss, g = ops.interlevel_loss_level(c, w, cp, wp, pulse_width, need_grad=True)
        ctx.save_for_backward(g)
        ctx.n_rays = wp.shape[0]
        return loss.mean()

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, go):
        (g,) = ctx.saved_tensors
        return None, None, None, g * (go / ctx.n_rays), None


class DistortionLossFn(torch.autograd.Function):
    """distortion_loss (model_components/losses.py:137-156): mean over rays of lossfun_distortion(c, w)."""

    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, c, w):
        loss, g = ops.distortion_loss_rays(c, w, need_grad=True)
        ctx.save_for_backward(g)
        ctx.n_rays = w.shape[0]
        return loss.mean()

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, go):
        (g,) = ctx.saved_tensors
        return None, g * (go / ctx.n_rays)


# ------------------------------------------------------------------------------------------------
# The training step as a handful of nodes (csrc/train_fused.hip): what the reference's orchestration spreads over
# RaySamples views, cats and elementwise ops between the kernels stays inside them.
class ProposalRoundFn(torch.autograd.Function):
    """One round of the training-mode proposal sampler as ONE node (ray_samplers.py:640-652 + models/neurad.py:396):
    density of the round's samples (S2) -> RaySamples.get_weights (S3) -> render_depth_simple of the round, all from the
    round's bin EDGES [R,S+1].  args: table, decoder_weight, spec, static_scale, origins, directions, pixel_area, edges
    -> weights [R,S], prop_depth [R,1]."""

    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, table, decoder_weight, spec, static_scale, origins, directions, pixel_area, edges):
        ctx.set_materialize_grads(False)
        ps = ops.ProposalSpec(spec, table, static_scale, decoder_weight)
        starts, ends = edges[:, :-1], edges[:, 1:]
        if any(ctx.needs_input_grad):
            dens, lf = ops.proposal_density_fwd(ps, origins, directions, pixel_area, starts, ends, save_features=True)
            ctx.ps = ps
            ctx.save_for_backward(origins, directions, pixel_area, edges, dens, lf)
        else:  # frozen between scheduled updates / eval: nothing to save
            dens = ops.proposal_density_fwd(ps, origins, directions, pixel_area, starts, ends)
        return ops.prop_weights_fwd(edges, dens)

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, gw, gdepth):
        o, d, a, edges, dens, lf = ctx.saved_tensors
        if gw is None and gdepth is None:
            return (None,) * 8
        gdens = ops.prop_weights_bwd(edges, dens, None if gw is None else gw.contiguous(), gdepth)
        starts, ends = edges[:, :-1], edges[:, 1:]
        gt = gdec = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:  # (frozen tables under a camera optimizer: rays only)
            gt, gdec = ops.proposal_density_bwd(ctx.ps, o, d, a, starts, ends, dens, gdens, level_features=lf)
            gt, gdec = _like_param(gt, ctx.ps.table.dtype), gdec.reshape(ctx.ps.decoder_weight.shape)
        go, gd = (None, None)
        if ctx.needs_input_grad[4] or ctx.needs_input_grad[5]:  # the rays moved with a camera optimizer
            go, gd = _ray_grads(ctx.needs_input_grad[4], ctx.needs_input_grad[5], ctx.ps.grid, ctx.ps.table,
                                ctx.ps.static_scale, o, d, a, starts, ends, _proposal_genc(dens, gdens, ctx.ps.decoder_weight))
        return gt, gdec, None, None, go, gd, None, None


class PropWeightsFn(torch.autograd.Function):
    """RaySamples.get_weights + render_depth_simple of a proposal round from its bin edges, for densities that come from
    elsewhere (a proposal field with dynamic actors: static density + actor overlay at operator level).
    args: edges [R,S+1], densities [R,S] -> weights [R,S], prop_depth [R,1]"""

    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, edges, densities):
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(edges, densities)
        return ops.prop_weights_fwd(edges, densities)

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, gw, gdepth):
        edges, dens = ctx.saved_tensors
        if gw is None and gdepth is None:
            return None, None
        return None, ops.prop_weights_bwd(edges, dens, None if gw is None else gw.contiguous(), gdepth)


class NffRenderTrainFn(torch.autograd.Function):
    """get_nff_outputs behind the sampler (models/neurad.py:373-395) for the static scene as ONE node: fused field forward
    (NeuRADField.forward) -> SigmoidDensity with the learnable beta -> render_weight_from_alpha -> accumulation, sky
    residual, features, depth -> appearance embedding written beside the features.

    beta = None: the density head (use_sdf = False: trunc_exp -> render_weight_from_density, models/neurad.py:718-723).
    args: table, spec, static_scale, beta (raw parameter), beta_min, origins, directions, pixel_area, edges [R,S+1] (last
    edge = sky distance), emb_weight | None, sensor_idx | None, times | None, (duration, n_per_sensor, temporal), order |
    None, ovr_row | None, ovr_rows | None, ovr_dirs | None, pair_idx | None (dynamic actors: the rows of the samples
    inside a box, ops.field_fwd_train), gw0, gb0, gw1, gb1, fw0, fb0, fw1, fb1, fw2, fb2
    -> features [R, 32 + A], depth [R,1], accumulation [R,1], weights of the non-sky samples [R,S-1]"""

    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, table, spec, static_scale, beta, beta_min, origins, directions, pixel_area, edges, emb_weight,
                sensor_idx, times, emb_cfg, order, ovr_row, ovr_rows, ovr_dirs, pair_idx, *params):
        ctx.set_materialize_grads(False)
        gw, gb, fw, fb = list(params[0:4:2]), list(params[1:4:2]), list(params[4:10:2]), list(params[5:10:2])
        fs = ops.FieldSpec(spec, table, static_scale, gw, gb, fw, fb, use_sdf=True, beta=1.0)  # (kernel head unused)
        override = None if ovr_rows is None else (ovr_row, ovr_rows, ovr_dirs)
        (feature, sdf, _head), (enc, hg, xf, hf) = ops.field_fwd_train(fs, origins, directions, pixel_area, edges[:, :-1],
                                                                      edges[:, 1:], order=order, override=override)
        ctx.has_ovr = ovr_rows is not None
        R, S = edges.shape[0], edges.shape[1] - 1
        A = 0 if emb_weight is None else emb_weight.shape[1]
        alpha, w_ns, out, depth, acc = ops.sdf_render_fwd(sdf.view(R, S), beta, beta_min, feature.view(R, S, -1), edges,
                                                          extra_cols=A)
        if A:
            ops.appearance_fwd(emb_weight, sensor_idx, times, emb_cfg[0], emb_cfg[1], emb_cfg[2], R, out=out[:, out.shape[1] - A:])
        ctx.spec, ctx.scale, ctx.table_dtype, ctx.beta_min, ctx.emb_cfg = spec, static_scale, table.dtype, beta_min, emb_cfg
        ctx.n_embed, ctx.A = (0 if emb_weight is None else emb_weight.shape[0]), A
        ctx.has = (sensor_idx is not None, times is not None)
        ctx.rays = ctx.needs_input_grad[5] or ctx.needs_input_grad[6]  # a camera optimizer moved the rays
        opt = ([t for t in (sensor_idx, times) if t is not None] + ([ovr_row, pair_idx] if ctx.has_ovr else [])
               + ([table] if ctx.rays else []))
        ctx.density_head = beta is None
        ctx.save_for_backward(origins, directions, pixel_area, edges, enc, hg, xf, hf, feature, sdf, alpha,
                              sdf.new_empty(0) if beta is None else beta, *params, *opt)
        return out, depth, acc, w_ns

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, g_out, g_depth, g_acc, g_wns):
        o, d, a, edges, enc, hg, xf, hf, feature, sdf, alpha, beta, *rest = ctx.saved_tensors
        params, opt = rest[:10], list(rest[10:])
        sensor_idx = opt.pop(0) if ctx.has[0] else None
        times = opt.pop(0) if ctx.has[1] else None
        R, S = edges.shape[0], edges.shape[1] - 1
        C_ = feature.shape[-1]
        if g_out is None:
            g_out = torch.zeros((R, C_ + ctx.A), device=o.device, dtype=torch.float32)
        gfeat, gsdf, gbeta = ops.sdf_render_bwd(sdf.view(R, S), None if ctx.density_head else beta, ctx.beta_min, alpha,
                                                feature.view(R, S, C_), edges, g_out[:, :C_], g_depth, g_acc, g_wns)
        g_emb = None
        if ctx.A and ctx.needs_input_grad[9]:
            g_emb = ops.appearance_bwd(g_out[:, C_:], sensor_idx, times, ctx.emb_cfg[0], ctx.emb_cfg[1], ctx.emb_cfg[2],
                                       ctx.n_embed)
        override = (opt.pop(0), opt.pop(0)) if ctx.has_ovr else None
        rays = (opt.pop(0), ctx.needs_input_grad[5], ctx.needs_input_grad[6]) if ctx.rays else None
        gt, grads, g_rows, (go, gd) = _field_backward(ctx.spec, ctx.scale, ctx.table_dtype, ctx.needs_input_grad[0], o, d, a,
                                                      edges[:, :-1], edges[:, 1:], enc, hg, xf, hf, params,
                                                      gfeat.view(R * S, C_), gsdf.view(-1), override=override, rays=rays)
        if not ctx.needs_input_grad[15]:
            g_rows = None
        g_beta = gbeta.reshape(beta.shape) if (ctx.needs_input_grad[3] and not ctx.density_head) else None
        return (gt, None, None, g_beta, None, go, gd, None, None, g_emb, None, None, None, None, None, g_rows, None, None,
                *grads)


class LidarLossFn(torch.autograd.Function):
    """The lidar terms of get_metrics_dict (models/neurad.py:485-521) in one launch each way.
    args: (non_return_distance, non_return_mult, quantile), lidar_rows int64 [n], inverse int32 [R] (ops.mask_compact),
    distance [n,1], did_return [n] bool, intensity_target [n,1], intensity [n,1], ray_drop_logits [n,1], depth [R,1],
    prop_depth_0 [R,1], ...  -> metrics [2 + n_levels]: depth_loss, intensity_loss, ray_drop_loss, depth_loss_0, ..."""

    @staticmethod
    def forward(ctx, cfg, lidar_rows, inverse, distance, did_return, intensity_target, intensity, logits, *depths):
        metrics, saved = ops.lidar_losses(depths, lidar_rows, distance, did_return, intensity, intensity_target, logits,
                                          cfg[0], cfg[1], cfg[2])
        ctx.save_for_backward(inverse, *saved)
        ctx.n_levels, ctx.n_rays = len(depths), depths[0].shape[0]
        return metrics

    @staticmethod
    def backward(ctx, g):
        inverse, *saved = ctx.saved_tensors
        need = ctx.needs_input_grad
        gds, gi, gl = ops.lidar_losses_bwd(saved, inverse, g.contiguous(), ctx.n_levels, ctx.n_rays, need[8:], need[6], need[7])
        return (None,) * 6 + (gi, gl, *gds)


class WeightedSumFn(torch.autograd.Function):
    """sum_i mult_i * term_i over 0-dim loss terms: the trainer's `functools.reduce(torch.add, loss_dict.values())` over
    `mult * metric` entries (engine/trainer.py:551) is two launches per term; this is one stack + one dot each way."""

    @staticmethod
    def forward(ctx, mults, *terms):
        ctx.save_for_backward(mults)
        return torch.dot(torch.stack([t.reshape(()) for t in terms]), mults)

    @staticmethod
    def backward(ctx, g):
        (mults,) = ctx.saved_tensors
        gs = (mults * g).unbind(0)
        return (None, *gs)

