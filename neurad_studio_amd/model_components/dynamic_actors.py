"""DynamicActors: actor trajectories as parameters/buffers with the reference's names
(mirror of nerfstudio/model_components/dynamic_actors.py:44-300, the part the hot path reads).

Pose interpolation, box tests and actor-grid lookups run in the HIP kernels (csrc/actors.hip); this module only
owns the state (so checkpoints and the trajectory optimiser's parameter group keep working)."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Tuple

import torch
from torch import Tensor, nn


@dataclass
class DynamicActorsConfig:  # dynamic_actors.py:31-41
    optimize_trajectories: bool = True
    actor_bbox_padding: Tuple[float, float, float] = (0.25, 0.25, 0.1)

    def setup(self, **kwargs):
        return DynamicActors(self, **kwargs)


def matrix_to_rotation_6d(matrix: Tensor) -> Tensor:  # cameras/camera_utils.py:446-464
    return matrix[..., :2, :].clone().reshape(*matrix.shape[:-2], 6)


def world2box_pairs(actors, query_times: Tensor, actor_idx: Tensor) -> Tuple[Tensor, Tensor]:
    """``actors``: this package's DynamicActors or the reference's (same parameter / buffer names,
    model_components/dynamic_actors.py:153-170).  Differentiable (w.r.t. actor_positions / actor_rotations_6d) world->box transform of ``actor_idx[m]`` at
    ``query_times[m]``: interpolate_trajectories_6d (utils/poses.py:90-150) + rotation_6d_to_matrix
    (cameras/camera_utils.py:422-443) + pose inverse (utils/poses.py:42-55), evaluated only for the (few)
    sample/actor pairs the HIP kernels reported as hits.  -> (R_inv [M,3,3], t_inv [M,3])

    The torch formulation of what csrc/actors.hip:actor_pair_positions_kernel and its hand-derived backward compute; the
    product path calls the kernels (autograd.ActorPairPositionsFn), the tests use this function to pin them."""
    F = torch.nn.functional
    poses = torch.cat([actors.actor_rotations_6d, actors.actor_positions], dim=-1)
    a1 = F.normalize(poses[..., :3], dim=-1)
    a2 = poses[..., 3:6]
    a2 = F.normalize(a2 - (a1 * a2).sum(-1, keepdim=True) * a1, dim=-1)
    poses = torch.cat([a1, a2, poses[..., 6:9]], dim=-1)
    ts = actors.unique_timestamps
    right = torch.searchsorted(ts, query_times.contiguous())
    left = (right - 1).clamp(min=0)
    right = right.clamp(max=len(ts) - 1)
    frac = ((query_times - ts[left]) / (ts[right] - ts[left] + 1e-6)).clamp(0.0, 1.0)
    pl, pr = poses[left, actor_idx], poses[right, actor_idx]
    ip = pl + (pr - pl) * frac[:, None]
    b1 = F.normalize(ip[:, :3], dim=-1)
    b2 = F.normalize(ip[:, 3:6] - (b1 * ip[:, 3:6]).sum(-1, keepdim=True) * b1, dim=-1)
    b3 = torch.cross(b1, b2, dim=-1)
    rot = torch.stack((b1, b2, b3), dim=-2)          # boxes2world rotation (rows b1,b2,b3)
    r_inv = rot.transpose(-2, -1)
    # (an elementwise 3-term dot: as a batched 3x3 GEMM over ~10^5 pairs this and its two backward products cost 1.1 ms
    #  per step in hipBLASLt kernels tiled for large matrices)
    t_inv = -(r_inv * ip[:, None, 6:]).sum(-1)
    return r_inv, t_inv


class DynamicActors(nn.Module):
    def __init__(self, config: DynamicActorsConfig, trajectories: List[dict]):
        super().__init__()
        self.config = config
        self._populate_actors(trajectories)
        self.requires_grad_(config.optimize_trajectories)
        # eval-time edit of the boxes (dynamic_actors.py:53-59, read by edit_boxes2world :181-249 outside training): the viewer's
        # sliders and the actor-shift evaluation write it; the kernels apply it after the pose interpolation
        # (nrhip_actor_prepare_edited)
        self.actor_editing = {"lateral": 0.0, "longitudinal": 0.0, "rotation": 0.0, "index": -1.0, "height": 0.0}

    def actor_bounds(self) -> Tensor:
        return self.actor_sizes / 2 + self.actor_padding

    def _populate_actors(self, trajectories: List[dict]) -> None:  # dynamic_actors.py:109-170
        uniq = torch.tensor(sorted({t.item() for traj in trajectories for t in traj["timestamps"]}), dtype=torch.float32)
        self.n_actors, self.n_times = len(trajectories), len(uniq)
        poses = torch.eye(4, dtype=torch.float32).view(1, 1, 4, 4).repeat(self.n_times, self.n_actors, 1, 1)
        present = torch.zeros((self.n_times, self.n_actors), dtype=torch.bool)
        sizes = torch.zeros((self.n_actors, 3), dtype=torch.float32)
        symmetric = torch.zeros((self.n_actors,), dtype=torch.bool)
        deformable = torch.zeros((self.n_actors,), dtype=torch.bool)
        for ai, traj in enumerate(trajectories):
            sizes[ai] = traj["dims"]
            symmetric[ai] = traj["symmetric"]
            deformable[ai] = traj["deformable"]
            for ti, t in enumerate(uniq):
                diff = (traj["timestamps"] - t).abs()
                k = diff.argmin(dim=0)
                if diff[k] < 1e-4:
                    present[ti, ai] = True
                poses[ti, ai] = traj["poses"][k]  # absent timestamps duplicate the closest pose (:143-149)
        self.register_buffer("unique_timestamps", uniq)
        self.register_buffer("actor_poses_at_time", poses)
        self.register_buffer("actor_present_at_time", present)
        self.register_buffer("actor_sizes", sizes)
        self.register_buffer("actor_symmetric", symmetric)
        self.register_buffer("actor_deformable", deformable)
        self.register_buffer("actor_padding", torch.tensor(self.config.actor_bbox_padding))
        self.register_buffer("actor_to_id", torch.arange(self.n_actors, dtype=torch.int64))
        self.actor_positions = nn.Parameter(poses[..., :3, 3].clone())
        self.actor_rotations_6d = nn.Parameter(matrix_to_rotation_6d(poses[..., :3, :3]))
        self.register_buffer("initial_positions", self.actor_positions.detach().clone())
        self.register_buffer("initial_rotations_6d", self.actor_rotations_6d.detach().clone())
        self.actor_vel_linear = nn.Parameter(torch.zeros((self.n_times, self.n_actors, 3)))
        self.actor_vel_angular = nn.Parameter(torch.zeros((self.n_times, self.n_actors, 3)))

    def world2box_pairs(self, query_times: Tensor, actor_idx: Tensor) -> Tuple[Tensor, Tensor]:
        return world2box_pairs(self, query_times, actor_idx)

    def requires_grad_(self, requires: bool = True):
        self.actor_positions.requires_grad_(requires)
        self.actor_rotations_6d.requires_grad_(requires)
        return self

    def get_param_groups(self, param_groups: Dict):
        if self.config.optimize_trajectories:
            param_groups["trajectory_opt"] = param_groups.get("trajectory_opt", []) + list(self.parameters())
