"""Samplers of the hot path with the reference's class names and call signatures
(nerfstudio/model_components/ray_samplers.py:32-132,255-376,569-666,838-852), running on HIP kernels.

Randomness (training-mode jitter) is drawn with torch on the device exactly where the reference draws it
and INJECTED into the kernels, so a seeded reference run can be reproduced sample for sample."""
from __future__ import annotations

import contextlib
from typing import Callable, List, Optional, Sequence, Tuple

import torch
from torch import Tensor, nn

from .. import ops
from ..cameras.rays import RayBundle, RaySamples


class PowerSpacing:
    """The ``spacing_to_euclidean_fn`` closure of SpacedSampler (ray_samplers.py:117-118) for the ZipNeRF power
    transform (utils/math.py:541-579), carrying what the PDF kernel needs to redo it on chip."""

    def __init__(self, nears: Tensor, fars: Tensor, lam: float, scaling: float):
        self.nears, self.fars, self.lam, self.scaling = nears, fars, lam, scaling

    def _fn(self, x):
        lam, lam_1 = self.lam, abs(self.lam - 1)
        return (lam_1 / lam) * ((x * self.scaling / lam_1 + 1) ** lam - 1)

    def __call__(self, x: Tensor) -> Tensor:
        lam, lam_1 = self.lam, abs(self.lam - 1)
        s_near, s_far = self._fn(self.nears), self._fn(self.fars)
        y = x * s_far + (1 - x) * s_near
        return (((y * lam / lam_1 + 1).clamp_min(1e-10) ** (1 / lam) - 1) * lam_1) / self.scaling


def _edges_to_samples(ray_bundle: RayBundle, sp: Tensor, eu: Tensor, fn) -> RaySamples:
    return ray_bundle.get_ray_samples(bin_starts=eu[..., :-1, None], bin_ends=eu[..., 1:, None],
                                      spacing_starts=sp[..., :-1, None], spacing_ends=sp[..., 1:, None],
                                      spacing_to_euclidean_fn=fn)


class Sampler(nn.Module):
    def __init__(self, num_samples: Optional[int] = None) -> None:
        super().__init__()
        self.num_samples = num_samples

    def forward(self, *args, **kwargs):
        return self.generate_ray_samples(*args, **kwargs)


class PowerSampler(Sampler):
    """ray_samplers.py:838-852 on top of SpacedSampler (:55-132)."""

    def __init__(self, num_samples: Optional[int] = None, lambda_: float = -1.5, scaling: float = 2.0,
                 train_stratified: bool = True, single_jitter: bool = False) -> None:
        super().__init__(num_samples)
        self.lambda_, self.scaling = lambda_, scaling
        self.train_stratified, self.single_jitter = train_stratified, single_jitter

    def generate_ray_samples(self, ray_bundle: Optional[RayBundle] = None, num_samples: Optional[int] = None) -> RaySamples:
        assert ray_bundle is not None and ray_bundle.nears is not None and ray_bundle.fars is not None
        num_samples = num_samples or self.num_samples
        assert num_samples is not None
        R = ray_bundle.origins.shape[0]
        t_rand = None
        if self.train_stratified and self.training:
            dev = ray_bundle.origins.device
            t_rand = (torch.rand((R, 1), device=dev).expand(R, num_samples + 1).contiguous() if self.single_jitter
                      else torch.rand((R, num_samples + 1), device=dev))
        sp, eu = ops.power_sampler(ray_bundle.nears, ray_bundle.fars, num_samples, self.lambda_, self.scaling, t_rand)
        return _edges_to_samples(ray_bundle, sp, eu, PowerSpacing(ray_bundle.nears, ray_bundle.fars, self.lambda_,
                                                                  self.scaling))


class PDFSampler(Sampler):
    """ray_samplers.py:255-376 (include_original=False, the only mode NeuRAD uses, :606)."""

    def __init__(self, num_samples: Optional[int] = None, train_stratified: bool = True, single_jitter: bool = False,
                 include_original: bool = False, histogram_padding: float = 0.01) -> None:
        super().__init__(num_samples)
        if include_original:
            raise NotImplementedError("include_original=True is not used by ProposalNetworkSampler (ray_samplers.py:606)")
        self.train_stratified, self.single_jitter = train_stratified, single_jitter
        self.histogram_padding = histogram_padding

    def generate_ray_samples(self, ray_bundle: Optional[RayBundle] = None, ray_samples: Optional[RaySamples] = None,
                             weights: Optional[Tensor] = None, num_samples: Optional[int] = None,
                             eps: float = 1e-5) -> RaySamples:
        if ray_samples is None or ray_bundle is None:
            raise ValueError("ray_samples and ray_bundle must be provided")
        assert weights is not None, "weights must be provided"
        num_samples = num_samples or self.num_samples
        fn = ray_samples.spacing_to_euclidean_fn
        if not isinstance(fn, PowerSpacing):
            raise NotImplementedError("the HIP PDF sampler resamples in ZipNeRF power spacing (PowerSampler bins)")
        assert ray_samples.spacing_starts is not None and ray_samples.spacing_ends is not None
        existing = torch.cat([ray_samples.spacing_starts[..., 0], ray_samples.spacing_ends[..., -1:, 0]], dim=-1)
        R = existing.shape[0]
        rand = None
        if self.train_stratified and self.training:
            rand = torch.rand((R,) if self.single_jitter else (R, num_samples + 1), device=existing.device)
        sp, eu = ops.pdf_sample(weights[..., 0].detach().contiguous(), existing.contiguous(), fn.nears, fn.fars,
                                num_samples, fn.lam, fn.scaling, self.histogram_padding, rand)
        return _edges_to_samples(ray_bundle, sp, eu, fn)  # bins are detached (ray_samplers.py:363-364)


class ProposalNetworkSampler(Sampler):
    """ray_samplers.py:569-666.  ``generate_ray_samples(ray_bundle, density_fns, pass_ray_samples)`` keeps the
    reference's orchestration (any callables work as density_fns); ``generate_fused`` runs the whole chain
    -- bins, densities, weights, resampling of every round -- as ONE kernel, one wavefront per ray."""

    def __init__(self, num_proposal_samples_per_ray: Tuple[int, ...] = (64,), num_nerf_samples_per_ray: int = 32,
                 num_proposal_network_iterations: int = 2, single_jitter: bool = False,
                 update_sched: Callable = lambda x: 1, initial_sampler: Optional[Sampler] = None,
                 pdf_sampler: Optional[PDFSampler] = None) -> None:
        super().__init__()
        self.num_proposal_samples_per_ray = num_proposal_samples_per_ray
        self.num_nerf_samples_per_ray = num_nerf_samples_per_ray
        self.num_proposal_network_iterations = num_proposal_network_iterations
        self.update_sched = update_sched
        if num_proposal_network_iterations < 1:
            raise ValueError("num_proposal_network_iterations must be >= 1")
        self.initial_sampler = initial_sampler if initial_sampler is not None else PowerSampler()
        self.pdf_sampler = pdf_sampler if pdf_sampler is not None else PDFSampler(include_original=False,
                                                                                  single_jitter=single_jitter)
        self._anneal, self._steps_since_update, self._step = 1.0, 0, 0

    def set_anneal(self, anneal: float) -> None:
        self._anneal = anneal

    def step_cb(self, step):
        self._step = step
        self._steps_since_update += 1

    def _proposals_train_this_step(self) -> bool:
        """the update schedule of the proposal networks (ray_samplers.py:630): every step for the first ten, then
        whenever more than update_sched(step) steps have passed since the last one"""
        return self._step < 10 or self._steps_since_update > self.update_sched(self._step)

    def generate_ray_samples(self, ray_bundle: Optional[RayBundle] = None, density_fns: Optional[List[Callable]] = None,
                             pass_ray_samples: bool = False) -> Tuple[RaySamples, List, List]:
        """Operator-level orchestration with the reference's contract (ray_samplers.py:614-666): round k evaluates
        density_fns[k] on its samples, turns the densities into weights and resamples the next round's bins from them
        (annealed); returns the field's samples plus every proposal round's (weights, samples)."""
        if ray_bundle is None or density_fns is None:
            raise ValueError("ray_bundle and density_fns must be provided")
        fns = list(density_fns) if pass_ray_samples else [
            (lambda rs, f=f: f(rs.frustums.get_positions())) for f in density_fns]
        rounds, train_props = self.num_proposal_network_iterations, self._proposals_train_this_step()
        counts = tuple(self.num_proposal_samples_per_ray[:rounds]) + (self.num_nerf_samples_per_ray,)
        samples = self.initial_sampler(ray_bundle, num_samples=counts[0])
        round_weights, round_samples = [], []
        for k in range(rounds):
            with contextlib.nullcontext() if train_props else torch.no_grad():  # frozen between scheduled updates
                density = fns[k](samples)
            w = samples.get_weights(density)
            round_weights.append(w)
            round_samples.append(samples)
            samples = self.pdf_sampler(ray_bundle, samples, w if self._anneal == 1.0 else w.pow(self._anneal),
                                       num_samples=counts[k + 1])
        if train_props:
            self._steps_since_update = 0
        return samples, round_weights, round_samples

    @torch.no_grad()
    def generate_fused(self, ray_bundle: RayBundle, proposal_fields: Sequence, sky_distance: float = 20000.0,
                       actor_cand=None):
        """Eval-mode fast path: proposal_fields[i] is the field evaluated in round i (pass the SAME field twice to
        reproduce the reference's late-binding quirk, models/neurad.py:248).  actor_cand: per-ray candidate lists
        (ops.actor_prepare) of a scene with dynamic actors -- the proposal fields' actor grids then take part."""
        if not isinstance(self.initial_sampler, PowerSampler):
            raise NotImplementedError("fused sampler needs PowerSampler bins")
        ns = tuple(self.num_proposal_samples_per_ray[: self.num_proposal_network_iterations]) + (
            self.num_nerf_samples_per_ray,)
        specs = [f.proposal_spec() for f in proposal_fields]
        lam, scaling = self.initial_sampler.lambda_, self.initial_sampler.scaling
        actor_specs = None if actor_cand is None else [f.hashgrid.actor_spec() for f in proposal_fields]
        ws, sps, eus = ops.proposal_sampler_fwd(specs, ray_bundle.origins, ray_bundle.directions, ray_bundle.pixel_area,
                                                ray_bundle.nears, ray_bundle.fars, ns, lam, scaling,
                                                self.pdf_sampler.histogram_padding, sky_distance, actor_specs, actor_cand)
        fars = ray_bundle.fars if ray_bundle.fars is not None else torch.full_like(ray_bundle.pixel_area, sky_distance)
        fn = PowerSpacing(ray_bundle.nears if ray_bundle.nears is not None else torch.zeros_like(fars),
                          fars.clamp_max(sky_distance), lam, scaling)
        rs_list = [_edges_to_samples(ray_bundle, sps[i], eus[i], fn) for i in range(len(ns))]
        return rs_list[-1], [w[..., None] for w in ws], rs_list[:-1]


class VolumetricSampler(Sampler):
    """ray_samplers.py:401-566: occupancy-grid march producing PACKED samples (ray_indices + [M] starts/ends).
    ``occupancy_grid`` is anything with nerfacc's ``OccGridEstimator.sampling`` signature
    (neurad_studio_amd.shims.nerfacc.OccGridEstimator runs the wavefront-compaction HIP kernel)."""

    def __init__(self, occupancy_grid, density_fn: Optional[Callable] = None, alpha_fn: Optional[Callable] = None):
        super().__init__()
        assert occupancy_grid is not None
        assert not (alpha_fn is not None and density_fn is not None), "density_fn and alpha_fn cannot be both set"
        self.occupancy_grid, self.density_fn, self.alpha_fn = occupancy_grid, density_fn, alpha_fn

    def _wrap(self, fn, origins, directions, times):
        if fn is None or not self.training:
            return None

        def wrapped(t_starts, t_ends, ray_indices):
            positions = origins[ray_indices] + directions[ray_indices] * (t_starts + t_ends)[:, None] / 2.0
            out = fn(positions) if times is None else fn(positions, times[ray_indices])
            return out.squeeze(-1)

        return wrapped

    def generate_ray_samples(self):
        raise RuntimeError("The VolumetricSampler fuses sample generation and density check together. "
                           "Please call forward() directly.")

    def forward(self, ray_bundle: RayBundle, render_step_size: float, near_plane: float = 0.0,
                far_plane: Optional[float] = None, alpha_thre: float = 0.01, cone_angle: float = 0.0):
        from ..cameras.rays import Frustums

        rays_o, rays_d = ray_bundle.origins.contiguous(), ray_bundle.directions.contiguous()
        t_min = t_max = None
        if ray_bundle.nears is not None and ray_bundle.fars is not None:
            t_min, t_max = ray_bundle.nears.contiguous().reshape(-1), ray_bundle.fars.contiguous().reshape(-1)
        ray_indices, starts, ends = self.occupancy_grid.sampling(
            rays_o=rays_o, rays_d=rays_d, t_min=t_min, t_max=t_max,
            sigma_fn=self._wrap(self.density_fn, rays_o, rays_d, ray_bundle.times),
            alpha_fn=self._wrap(self.alpha_fn, rays_o, rays_d, ray_bundle.times), render_step_size=render_step_size,
            near_plane=near_plane, far_plane=1e10 if far_plane is None else far_plane, stratified=self.training,
            cone_angle=cone_angle, alpha_thre=alpha_thre)
        if starts.shape[0] == 0:  # single fake sample (ray_samplers.py:541-547)
            ray_indices = torch.zeros((1,), dtype=torch.long, device=rays_o.device)
            starts = torch.ones((1,), dtype=torch.float32, device=rays_o.device)
            ends = torch.ones((1,), dtype=torch.float32, device=rays_o.device)
        ray_samples = RaySamples(frustums=Frustums(origins=rays_o[ray_indices], directions=rays_d[ray_indices],
                                                   starts=starts[..., None], ends=ends[..., None],
                                                   pixel_area=ray_bundle.pixel_area[ray_indices]),
                                 camera_indices=None if ray_bundle.camera_indices is None
                                 else ray_bundle.camera_indices[ray_indices])
        if ray_bundle.times is not None:
            ray_samples.times = ray_bundle.times[ray_indices]
        return ray_samples, ray_indices
