"""The lidar supervision terms NeuRAD puts on the hot path's outputs every training step:
depth / intensity / ray-drop / carving losses for the final samples and depth / carving for each proposal round
(nerfstudio/models/neurad.py:485-521 in get_metrics_dict, weighted at :534-560 in get_loss_dict).

Inputs are exactly what ``get_nff_outputs(..., calc_lidar_losses=True)`` returns plus the decoded lidar head and the
lidar part of the batch.  A handful of elementwise GPU ops over the n_lidar rays -- the work is in the kernels that
produce the depths and weights; the sampler-side terms (interlevel, distortion) are HIP kernels in ``losses.py``."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict

import torch
from torch import Tensor
from torch.nn.functional import binary_cross_entropy_with_logits


@dataclass
class LidarLossSettings:
    """the lidar-related members of the reference's LossSettings (models/neurad.py:65-94), same names and defaults"""

    depth_mult: float = 0.01
    intensity_mult: float = 0.1
    carving_mult: float = 0.01
    quantile_threshold: float = 0.95
    non_return_lidar_distance: float = 150.0
    non_return_loss_mult: float = 0.1
    ray_drop_loss_mult: float = 0.01
    prop_lidar_loss_mult: float = 0.1


def _depth_l1(pred: Tensor, measured: Tensor, returned: Tensor, cfg: LidarLossSettings) -> Tensor:
    """per-ray |target - pred|: beams without a return are pulled beyond non_return_lidar_distance (never closer than
    where they already are) and down-weighted (models/neurad.py:491-495,513-517)"""
    beyond = pred.detach().clamp_min(cfg.non_return_lidar_distance)
    err = (torch.where(returned[:, None], measured, beyond) - pred).abs()
    return torch.where(returned[:, None], err, err * cfg.non_return_loss_mult)


def lidar_rows(is_lidar: Tensor, n_lidar: int):
    """(rows int64 [n_lidar], inverse int32 [R]) of the lidar rays of a batch without the host sync of ``is_lidar.nonzero()``
    (the caller knows n_lidar: the lidar part of the batch comes with the batch)"""
    from .. import ops

    return ops.mask_compact(is_lidar, n_lidar)


def lidar_metrics(outputs: Dict[str, Tensor], is_lidar: Tensor, did_return: Tensor, distance: Tensor,
                  intensity_target: Tensor, cfg: LidarLossSettings, num_proposal_rounds: int = 2, fused: bool = True,
                  rows=None) -> Dict[str, Tensor]:
    """is_lidar [R] bool over the whole batch; did_return [n_lidar] bool, distance [n_lidar,1], intensity_target
    [n_lidar,1] for the lidar rays in batch order.  outputs: depth [R,1], prop_depth_i [R,1], prop_weights_loss_i,
    non_nearby_weights (or non_nearby_weights_loss, its squared sum), and the lidar head's intensity / ray_drop_logits
    [n_lidar,1].  fused: the depth / intensity / ray-drop terms in one launch each way (nrhip_lidar_losses: the quantile is
    a radix select); False: the same terms as torch ops (A/B, debugging).  rows: ``lidar_rows(is_lidar, n_lidar)`` when the
    caller already has it."""
    n_lidar = distance.shape[0]  # (== is_lidar.sum(): the lidar part of the batch comes with the batch)
    if "non_nearby_weights_loss" in outputs:
        carving = outputs["non_nearby_weights_loss"] / n_lidar
    else:
        carving = outputs["non_nearby_weights"].square().sum() / n_lidar  # average per lidar ray
    if fused:
        from .. import autograd as ag

        rows, inverse = rows if rows is not None else lidar_rows(is_lidar, n_lidar)
        v = ag.LidarLossFn.apply((cfg.non_return_lidar_distance, cfg.non_return_loss_mult, cfg.quantile_threshold), rows,
                                 inverse, distance, did_return, intensity_target, outputs["intensity"],
                                 outputs["ray_drop_logits"], outputs["depth"],
                                 *[outputs[f"prop_depth_{i}"] for i in range(num_proposal_rounds)]).unbind(0)
        m = {"depth_loss": v[0], "intensity_loss": v[1], "ray_drop_loss": v[2], "carving_loss": carving}
        for i in range(num_proposal_rounds):
            m[f"depth_loss_{i}"] = v[3 + i]
            m[f"carving_loss_{i}"] = outputs[f"prop_weights_loss_{i}"] / n_lidar
        return m
    # One compaction of the lidar rays for all three depth terms, and masked means instead of boolean indexing: every
    # `x[mask]` is a nonzero + a device->host sync (its length) -- five of them per step in the reference formulation.
    lidar_idx = is_lidar.reshape(-1).nonzero().squeeze(-1)
    err = _depth_l1(outputs["depth"].index_select(0, lidar_idx), distance, did_return, cfg)
    # robust mean: the worst (1 - quantile_threshold) of the rays are left out of the depth and intensity terms
    keep = (err < torch.quantile(err, cfg.quantile_threshold)).squeeze(-1)
    m = {"depth_loss": (err.squeeze(-1) * keep).sum() / keep.sum()}
    sel = keep & did_return
    m["intensity_loss"] = ((intensity_target - outputs["intensity"]).square().squeeze(-1) * sel).sum() / sel.sum()
    logits = outputs["ray_drop_logits"]
    m["ray_drop_loss"] = binary_cross_entropy_with_logits(logits, (~did_return)[:, None].to(logits))
    m["carving_loss"] = carving
    for i in range(num_proposal_rounds):
        m[f"depth_loss_{i}"] = _depth_l1(outputs[f"prop_depth_{i}"].index_select(0, lidar_idx), distance, did_return,
                                         cfg).mean()
        m[f"carving_loss_{i}"] = outputs[f"prop_weights_loss_{i}"] / n_lidar
    return m


def lidar_loss_multipliers(cfg: LidarLossSettings, num_proposal_rounds: int = 2) -> Dict[str, float]:
    """metric name -> multiplier of get_loss_dict (models/neurad.py:534-560)"""
    out = {"depth_loss": cfg.depth_mult, "intensity_loss": cfg.intensity_mult, "carving_loss": cfg.carving_mult,
           "ray_drop_loss": cfg.ray_drop_loss_mult}
    for i in range(num_proposal_rounds):
        out[f"depth_loss_{i}"] = cfg.prop_lidar_loss_mult * cfg.depth_mult
        out[f"carving_loss_{i}"] = cfg.prop_lidar_loss_mult * cfg.carving_mult
    return out


class WeightedLossSum:
    """total = sum_k mult_k * term_k as ONE stack + ONE dot (autograd.WeightedSumFn) instead of a multiply and an add per
    term; the multipliers live on the device once.  ``terms``: name -> 0-dim tensor, ``mults``: name -> float."""

    def __init__(self, mults: Dict[str, float], device) -> None:
        self.names = list(mults)
        self.mults = torch.tensor([float(mults[k]) for k in self.names], dtype=torch.float32, device=device)

    def __call__(self, terms: Dict[str, Tensor]) -> Tensor:
        from .. import autograd as ag

        return ag.WeightedSumFn.apply(self.mults, *[terms[k] for k in self.names])


def lidar_loss_dict(metrics: Dict[str, Tensor], cfg: LidarLossSettings, num_proposal_rounds: int = 2) -> Dict[str, Tensor]:
    out = {"depth_loss": cfg.depth_mult * metrics["depth_loss"],
           "intensity_loss": cfg.intensity_mult * metrics["intensity_loss"],
           "carving_loss": cfg.carving_mult * metrics["carving_loss"],
           "ray_drop_loss": cfg.ray_drop_loss_mult * metrics["ray_drop_loss"]}
    for i in range(num_proposal_rounds):
        out[f"depth_loss_{i}"] = cfg.prop_lidar_loss_mult * cfg.depth_mult * metrics[f"depth_loss_{i}"]
        out[f"carving_loss_{i}"] = cfg.prop_lidar_loss_mult * cfg.carving_mult * metrics[f"carving_loss_{i}"]
    return out
