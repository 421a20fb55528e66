"""The drop-in boundary against the REFERENCE's own objects (CPU, build container only: skipped where /root/reference
is absent, e.g. on the GPU box).

  * operator level: the reference's HashEncoding / MLP / MLPWithHashEncoding / SHEncoding(implementation="tcnn") and
    NeuRADField / NeuRADProposalField construct on the ``tinycudann`` import-name package with the parameter names of
    SURVEY App. B (subprocess: the torch-path tests of this process need tinycudann ABSENT);
  * module level: the reference's RayBundle / RaySamples / FieldHeadNames flow through this package's sampler and
    field down to the ``ops`` calls (ops replaced by the CPU oracle -- there is no GPU here), and the ``neurad-hip``
    MethodSpecification builds the reference's NeuRADModel subclass whose eval outputs match the reference's own
    torch model on the same weights.
"""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import neurad_oracle as O  # noqa: E402
import ref_import  # noqa: E402
from conftest import rel_l2  # noqa: E402

pytestmark = pytest.mark.skipif(not ref_import.reference_available(), reason="reference tree not present")


def T(a):
    return torch.from_numpy(np.ascontiguousarray(np.asarray(a, np.float32)))


def N(t):
    return t.detach().cpu().numpy()


# ---- operator level ---------------------------------------------------------------------------------------------
def test_reference_tcnn_components_construct_on_the_shim():
    code = textwrap.dedent(f"""
        import sys
        sys.path[:0] = [{ROOT!r}, {os.path.join(ROOT, "oracle")!r}, {os.path.join(ROOT, "neurad_studio_amd", "integration")!r}]
        import tinycudann, nerfacc                     # the import-name packages
        assert tinycudann.__file__.startswith({ROOT!r}) and nerfacc.__version__ == "0.5.2"
        import ref_import
        ref_import.install(allow_tcnn=True)  # (same harness, tinycudann now present)
        from nerfstudio.utils.external import TCNN_EXISTS
        assert TCNN_EXISTS
        from nerfstudio.field_components.encodings import HashEncoding, SHEncoding
        from nerfstudio.field_components.mlp import MLP, MLPWithHashEncoding
        h = HashEncoding(num_levels=4, min_res=16, max_res=128, log2_hashmap_size=8, features_per_level=2, implementation="tcnn")
        assert [n for n, _ in h.named_parameters()] == ["tcnn_encoding.params"] and h.tcnn_encoding.params.numel() == 4 * 256 * 2
        m = MLP(in_dim=32, num_layers=2, layer_width=32, out_dim=33, implementation="tcnn")
        assert sorted(n for n, _ in m.named_parameters())[0].startswith("tcnn_encoding.layers.")
        mh = MLPWithHashEncoding(num_levels=4, min_res=16, max_res=128, log2_hashmap_size=8, features_per_level=2,
                                 num_layers=2, layer_width=32, out_dim=4, implementation="tcnn")
        names = [n for n, _ in mh.named_parameters()]
        assert "model.encoding.params" in names and any(n.startswith("model.network.layers.") for n in names), names
        s = SHEncoding(levels=4, implementation="tcnn")
        assert s.get_out_dim() == 16
        from nerfstudio.field_components.neurad_encoding import NeuRADHashEncodingConfig, ActorSettings, StaticSettings
        from nerfstudio.fields.neurad_field import NeuRADField, NeuRADFieldConfig, NeuRADProposalField, NeuRADProposalFieldConfig
        from nerfstudio.model_components.dynamic_actors import DynamicActors, DynamicActorsConfig
        grid = NeuRADHashEncodingConfig(static=StaticSettings(log2_hashmap_size=9),
                                        actor=ActorSettings(log2_hashmap_size=8, use_4d_hashgrid=False))
        actors = DynamicActors(DynamicActorsConfig(), trajectories=[])
        f = NeuRADField(NeuRADFieldConfig(grid=grid), actors=actors, static_scale=100.0, implementation="tcnn")
        names = [n for n, _ in f.named_parameters()]
        assert "hashgrid.static_grid.tcnn_encoding.params" in names and "sdf_to_density.beta" in names, names
        assert any(n.startswith("mlp_geo.tcnn_encoding.layers.") for n in names), names
        pcfg = NeuRADProposalFieldConfig()
        pcfg.grid.static.log2_hashmap_size = 9
        pcfg.grid.actor.use_4d_hashgrid = False
        p = NeuRADProposalField(pcfg, actors=actors, static_scale=100.0, implementation="tcnn")
        assert "density_decoder.weight" in [n for n, _ in p.named_parameters()]
        print("SHIM-OK")
        """)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, PYTHONDONTWRITEBYTECODE="1"))
    assert "SHIM-OK" in r.stdout, r.stderr[-3000:]


# ---- module level -----------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def ref():
    ref_import.install()
    import nerfstudio  # noqa: F401

    return True


class OracleOps:
    """CPU stand-in for the device: the ops entry points the module layer calls, answered by the oracle.  Records the
    shapes it was called with so the test can check what reached the boundary."""

    def __init__(self):
        self.calls = []

    def install(self, monkeypatch):
        from neurad_studio_amd import ops

        for name in ("power_sampler", "field_fwd", "proposal_sampler_fwd", "render_fwd", "proposal_density_fwd",
                     "accumulate_along_rays"):
            monkeypatch.setattr(ops, name, getattr(self, name))

    def accumulate_along_rays(self, weights, values=None):
        return T(O.accumulate_along_rays(N(weights), None if values is None else N(values)))

    def power_sampler(self, nears, fars, num_samples, lam=-1.0, scaling=0.1, t_rand=None, last_edge=0.0):
        self.calls.append(("power_sampler", tuple(fars.shape), num_samples))
        n = np.zeros(fars.numel(), np.float32) if nears is None else N(nears).reshape(-1)
        bins, eu, _ = O.power_sampler(n, N(fars).reshape(-1), num_samples, lam, scaling,
                                      None if t_rand is None else N(t_rand))
        return T(bins), T(eu)

    @staticmethod
    def _field_params(fs):
        g = fs.grid
        grid = O.GridParams(N(fs.table), g.num_levels, g.min_res, g.max_res, g.log2_hashmap_size)
        return O.FieldParams(grid, fs.static_scale, [N(w) for w in fs.geo_w], [N(b) for b in fs.geo_b],
                             [N(w) for w in fs.feat_w], [N(b) for b in fs.feat_b], beta=fs.beta, beta_min=0.0,
                             use_sdf=fs.use_sdf)

    def field_fwd(self, fs, origins, directions, pixel_area, starts, ends, order=None):
        self.calls.append(("field_fwd", tuple(origins.shape), tuple(directions.shape), tuple(pixel_area.shape),
                           tuple(starts.shape)))
        out = O.field_fwd(self._field_params(fs), N(origins), N(directions), N(pixel_area), N(starts), N(ends))
        head = out["alpha"] if fs.use_sdf else out["density"]
        return T(out["feature"]), T(out.get("sdf", np.log(np.maximum(head, 1e-30)))), T(head)

    def proposal_density_fwd(self, ps, origins, directions, pixel_area, starts, ends, save_features=False):
        self.calls.append(("proposal_density_fwd", tuple(starts.shape)))
        g = ps.grid
        p = O.ProposalParams(O.GridParams(N(ps.table), g.num_levels, g.min_res, g.max_res, g.log2_hashmap_size),
                             ps.static_scale, N(ps.decoder_weight))
        dens = T(O.proposal_density(p, N(origins), N(directions), N(pixel_area), N(starts), N(ends)))
        return (dens, torch.zeros(g.num_levels, dens.numel())) if save_features else dens

    def proposal_sampler_fwd(self, props, origins, directions, pixel_area, nears, fars, num_samples=(128, 64, 32),
                             lam=-1.0, scaling=0.1, histogram_padding=0.01, sky_distance=20000.0, actor_specs=None,
                             cand=None):
        assert actor_specs is None and cand is None, "the stub covers the static scene"
        self.calls.append(("proposal_sampler_fwd", tuple(origins.shape), tuple(num_samples)))
        pp = [O.ProposalParams(O.GridParams(N(p.table), p.grid.num_levels, p.grid.min_res, p.grid.max_res,
                                            p.grid.log2_hashmap_size), p.static_scale, N(p.decoder_weight)) for p in props]
        R = origins.shape[0]
        so = O.proposal_sampler(pp, N(origins), N(directions), N(pixel_area).reshape(-1),
                                np.zeros(R, np.float32) if nears is None else N(nears).reshape(-1),
                                N(fars).reshape(-1), tuple(num_samples[:-1]), num_samples[-1], lam, scaling, sky_distance,
                                late_binding_quirk=False, stretch_sky=False)
        edges = lambda s, e: T(np.concatenate([s, e[:, -1:]], -1))  # noqa: E731
        sps = [T(b) for b in so.prop_spacing] + [edges(so.spacing_starts, so.spacing_ends)]
        eus = [edges(s, e) for s, e in zip(so.prop_starts, so.prop_ends)] + [edges(so.starts, so.ends)]
        return [T(w) for w in so.prop_weights], sps, eus

    # ---- the entry points of the fused TRAINING nodes (forward only: the CPU stand-in has no backward) ----------------
    def install_training(self, monkeypatch):
        from neurad_studio_amd import ops

        for name in ("prop_weights_fwd", "pdf_sample", "field_fwd_train", "sdf_render_fwd", "appearance_fwd",
                     "lidar_carving"):
            monkeypatch.setattr(ops, name, getattr(self, name))

    def prop_weights_fwd(self, edges, densities, want_depth=True):
        self.calls.append(("prop_weights_fwd", tuple(densities.shape)))
        e, dn = N(edges), N(densities)
        w = O.weights_from_density(e[:, 1:] - e[:, :-1], dn)
        return T(w), T((w * (e[:, 1:] + e[:, :-1]) / 2).sum(-1, keepdims=True).astype(np.float32))

    def pdf_sample(self, weights, spacing_bins, nears, fars, num_samples, lam=-1.0, scaling=0.1, histogram_padding=0.01,
                   rand=None):
        self.calls.append(("pdf_sample", tuple(weights.shape), num_samples))
        R = weights.shape[0]
        n = np.zeros((R, 1), np.float32) if nears is None else N(nears).reshape(-1, 1)
        f = N(fars).reshape(-1, 1)
        sp = O.Spacing(O.power_fn(n * np.float32(scaling), lam), O.power_fn(f * np.float32(scaling), lam), lam, scaling)
        bins, eu = O.pdf_sample(N(weights), N(spacing_bins), num_samples, sp, histogram_padding,
                                rand=None if rand is None else N(rand).reshape(R, -1))
        return T(bins), T(eu)

    def field_fwd_train(self, fs, origins, directions, pixel_area, starts, ends, order=None, override=None):
        assert override is None, "the stub covers the static scene"
        self.calls.append(("field_fwd_train", tuple(starts.shape)))
        out = O.field_fwd(self._field_params(fs), N(origins), N(directions), N(pixel_area), N(starts), N(ends))
        n = starts.shape[0] * starts.shape[1]
        z = torch.zeros(n, 1)
        return (T(out["feature"]).reshape(n, -1), T(out["sdf"]).reshape(n), z[:, 0]), (z, z, z, z)

    def sdf_render_fwd(self, sdf, beta, beta_min, features, edges, extra_cols=0):
        self.calls.append(("sdf_render_fwd", tuple(sdf.shape)))
        e = N(edges)
        alpha = O.sigmoid(-N(sdf) * np.float32(abs(float(beta)) + beta_min))
        w, _ = O.render_weight_from_alpha(alpha)
        feats, depth, acc = O.composite(w, N(features), e[:, :-1], e[:, 1:])
        out = torch.zeros(sdf.shape[0], feats.shape[1] + extra_cols)
        out[:, :feats.shape[1]] = T(feats)
        return T(alpha), T(w[:, :-1].copy()), out, T(depth).reshape(-1, 1), T(acc).reshape(-1, 1)

    def appearance_fwd(self, weight, sensor_idx, times, duration, n_per_sensor, temporal, n_rays, out=None):
        self.calls.append(("appearance_fwd", n_rays))
        wt = N(weight)
        s = np.zeros(n_rays, np.int64) if sensor_idx is None else N(sensor_idx).reshape(-1)
        if temporal and times is not None:
            ti = N(times).reshape(-1) / np.float32(duration) * np.float32(n_per_sensor)
            lo = np.clip(np.floor(ti), 0, n_per_sensor - 1)
            hi = np.clip(lo + 1, 0, n_per_sensor - 1)
            fr = (ti - lo)[:, None].astype(np.float32)
            val = wt[(lo + s * n_per_sensor).astype(np.int64)] * (1 - fr) + wt[(hi + s * n_per_sensor).astype(np.int64)] * fr
        else:
            val = wt[s]
        out.copy_(T(val.astype(np.float32)))
        return out

    def lidar_carving(self, starts, ends, is_lidar, did_return, distance, carving_epsilon, non_return_lidar_distance,
                      weights=None, want_mask=True, want_grad=True):
        self.calls.append(("lidar_carving", tuple(starts.shape)))
        mid = (N(starts) + N(ends)) * 0.5
        lid = N(is_lidar).reshape(-1, 1).astype(bool)
        dist = N(distance).reshape(-1, 1)
        close_hit = np.abs(dist - mid) < carving_epsilon
        if did_return is None:
            close = lid & close_hit
        else:
            ret = N(did_return).reshape(-1, 1).astype(bool)
            close = lid & ((ret & close_hit) | (~ret & (mid < non_return_lidar_distance)))
        loss = gw = None
        if weights is not None:
            m = (lid & ~close).astype(np.float32)
            loss = T(((N(weights) * m) ** 2).sum(-1).astype(np.float32))
            gw = T((2 * N(weights) * m).astype(np.float32))
        return (torch.from_numpy(close) if want_mask else None), loss, gw

    def render_fwd(self, fs, origins, directions, pixel_area, starts, ends, return_weights=False, out=None,
                   early_stop_eps=0.0, order=None):
        self.calls.append(("render_fwd", tuple(origins.shape), tuple(starts.shape)))
        r = O.render_rays(self._field_params(fs), N(origins), N(directions), N(pixel_area).reshape(-1), N(starts), N(ends))
        res = (T(r["features"]), T(r["depth"]).reshape(-1, 1), T(r["accumulation"]).reshape(-1, 1))
        return res + (T(r["weights"]),) if return_weights else res


def _rays(R, seed=3, lidar_from=None):
    g = np.random.default_rng(seed)
    o = (g.normal(size=(R, 3)) * 5).astype(np.float32)
    d = g.normal(size=(R, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    return o, d


def test_reference_raysamples_and_headnames_flow_through_the_hip_field(ref, monkeypatch):
    from nerfstudio.cameras.rays import RayBundle as RefRayBundle
    from nerfstudio.cameras.rays import RaySamples as RefRaySamples
    from nerfstudio.field_components.field_heads import FieldHeadNames as RefHeads
    from nerfstudio.field_components.neurad_encoding import NeuRADHashEncodingConfig, StaticSettings
    from nerfstudio.fields.neurad_field import NeuRADField as RefField
    from nerfstudio.fields.neurad_field import NeuRADFieldConfig as RefFieldConfig

    from neurad_studio_amd.field_components.field_heads import FieldHeadNames
    from neurad_studio_amd.fields.neurad_field import NeuRADField
    from neurad_studio_amd.model_components.ray_samplers import PowerSampler

    assert FieldHeadNames is RefHeads  # the reference's own enum, not a look-alike
    dev = OracleOps()
    dev.install(monkeypatch)
    R, S = 24, 16
    o, d = _rays(R)
    rb = RefRayBundle(origins=T(o), directions=T(d), pixel_area=torch.full((R, 1), 2.43e-6), nears=torch.zeros(R, 1),
                      fars=torch.full((R, 1), 80.0), times=torch.rand(R, 1))
    rs = PowerSampler(num_samples=S, lambda_=-1.0, scaling=0.1).eval()(rb)
    assert isinstance(rs, RefRaySamples) and rs.frustums.origins.shape == (R, S, 3)  # the reference's broadcast views
    # the reference's config object (its grid._target is the reference encoding) drives the HIP field
    cfg = RefFieldConfig(grid=NeuRADHashEncodingConfig(static=StaticSettings(log2_hashmap_size=10)))
    torch.manual_seed(1)
    from nerfstudio.model_components.dynamic_actors import DynamicActors, DynamicActorsConfig

    actors = DynamicActors(DynamicActorsConfig(), trajectories=[])  # the reference's actor container, empty scene
    hip = NeuRADField(cfg, actors=actors, static_scale=torch.tensor(100.0), implementation="tcnn").eval()
    refm = RefField(cfg, actors=actors, static_scale=100.0, implementation="torch").eval()
    assert sorted(hip.state_dict()) == sorted(refm.state_dict())  # checkpoint-compatible names
    hip.hashgrid.static_grid.hash_table.data.mul_(500.0)
    refm.load_state_dict(hip.state_dict())
    with torch.no_grad():
        out = hip(rs)
        want = refm(rs)
    assert ("field_fwd", (R, 3), (R, 3), (R,), (R, S)) in dev.calls  # per-ray o/d/area + [R,S] edges reached the ABI layer
    assert set(out) == set(want) == {RefHeads.FEATURE, RefHeads.SDF, RefHeads.ALPHA}
    for k in want:
        assert out[k].shape == want[k].shape
        assert rel_l2(N(out[k]), N(want[k])) < 1e-5, k


def _tiny_nerfacc():
    """dense-mode nerfacc 0.5.2 maths for the REFERENCE model on CPU (it imports a stubbed nerfacc here)"""
    import types

    m = types.ModuleType("nerfacc")

    def render_weight_from_alpha(alphas, **kw):
        trans = torch.cumprod(torch.cat([torch.ones_like(alphas[..., :1]), 1 - alphas[..., :-1]], -1), -1)
        return trans * alphas, trans

    def accumulate_along_rays(weights, values=None, ray_indices=None, n_rays=None):
        return weights.sum(-1, keepdim=True) if values is None else (weights[..., None] * values).sum(-2)

    m.render_weight_from_alpha, m.accumulate_along_rays = render_weight_from_alpha, accumulate_along_rays
    return m


def test_neurad_hip_method_builds_the_reference_model_and_matches_its_torch_eval(ref, monkeypatch):
    import nerfstudio.model_components.renderers as ref_renderers
    import nerfstudio.models.neurad as ref_neurad
    from nerfstudio.cameras.rays import RayBundle as RefRayBundle
    from nerfstudio.data.scene_box import SceneBox
    from nerfstudio.plugins.types import MethodSpecification

    monkeypatch.setattr(ref_neurad, "VGGPerceptualLossPix2Pix", torch.nn.Identity)  # needs torchvision weights
    from neurad_studio_amd.integration.neurad_hip import NeuRADHipModel, neurad_hip

    assert isinstance(neurad_hip, MethodSpecification) and neurad_hip.config.method_name == "neurad-hip"
    mcfg = neurad_hip.config.pipeline.model
    assert mcfg.field.grid.actor.use_4d_hashgrid is False and mcfg.eval_num_rays_per_chunk == 1 << 15
    # registry path (plugins/registry.py:56-73): the env-var form resolves to the same object
    monkeypatch.setenv("NERFSTUDIO_METHOD_CONFIGS", "neurad-hip=neurad_studio_amd.integration.neurad_hip:neurad_hip")
    from nerfstudio.plugins.registry import discover_methods

    methods, descr = discover_methods()
    assert methods["neurad-hip"] is neurad_hip.config and "HIP" in descr["neurad-hip"]

    def shrink(c):
        c.field.grid.static.log2_hashmap_size = 10
        for pf in (c.sampling.proposal_field_1, c.sampling.proposal_field_2):
            pf.grid.static.log2_hashmap_size = 9
        c.loss.vgg_mult = 0.0
        return c

    from copy import deepcopy

    kw = dict(scene_box=SceneBox(aabb=torch.tensor([[-100.0] * 3, [100.0] * 3])), num_train_data=2,
              metadata={"duration": 8.0, "sensor_idx_to_name": {0: "cam", 1: "lidar"}, "trajectories": []})
    torch.manual_seed(0)
    hip = shrink(deepcopy(mcfg)).setup(**kw).eval()
    assert isinstance(hip, NeuRADHipModel) and isinstance(hip, ref_neurad.NeuRADModel)
    from neurad_studio_amd.fields.neurad_field import NeuRADField, NeuRADProposalField
    from neurad_studio_amd.model_components.ray_samplers import ProposalNetworkSampler

    assert isinstance(hip.field, NeuRADField) and all(isinstance(p, NeuRADProposalField) for p in hip.proposal_fields)
    assert isinstance(hip.sampler, ProposalNetworkSampler)
    ref_cfg = shrink(ref_neurad.NeuRADModelConfig(implementation="torch"))
    for c in (ref_cfg.field, ref_cfg.sampling.proposal_field_1, ref_cfg.sampling.proposal_field_2):
        c.grid.actor.use_4d_hashgrid = False
    refm = ref_cfg.setup(**kw).eval()
    assert sorted(hip.state_dict()) == sorted(refm.state_dict())  # a neurad checkpoint loads into neurad-hip
    with torch.no_grad():
        hip.field.hashgrid.static_grid.hash_table.mul_(500.0)
        for p in hip.proposal_fields:
            p.hashgrid.static_grid.hash_table.mul_(2000.0)
    refm.load_state_dict(hip.state_dict())
    # the reference on CPU: real compositing instead of its 0.5 placeholder (models/neurad.py:713-715)
    na = _tiny_nerfacc()
    monkeypatch.setattr(ref_neurad, "nerfacc", na)
    monkeypatch.setattr(ref_renderers, "nerfacc", na)
    monkeypatch.setattr(type(refm), "_render_weights",
                        lambda self, outputs, rs: na.render_weight_from_alpha(
                            outputs[ref_neurad.FieldHeadNames.ALPHA].squeeze(-1))[0])
    dev = OracleOps()
    dev.install(monkeypatch)
    R = 16
    o, d = _rays(R, seed=5)

    def bundle():
        return RefRayBundle(origins=T(o), directions=T(d), pixel_area=torch.full((R, 1), 2.7e-7), nears=None,
                            fars=None, times=torch.full((R, 1), 3.3),
                            metadata={"sensor_idxs": torch.zeros(R, 1, dtype=torch.long)})

    with torch.no_grad():
        got = hip.get_nff_outputs(bundle())
        want = refm.get_nff_outputs(bundle())
    assert [c[0] for c in dev.calls] == ["proposal_sampler_fwd", "render_fwd"]  # eval chunk = the two fused entry points
    assert dev.calls[0][2] == (128, 64, 32) and dev.calls[1][2] == (R, 32)
    assert set(got) == set(want)
    for k in ("features", "depth", "accumulation", "prop_depth_0", "prop_depth_1"):
        assert got[k].shape == want[k].shape, k
        assert rel_l2(N(got[k]), N(want[k])) < 2e-4, (k, rel_l2(N(got[k]), N(want[k])))
    assert got["features"].shape == (R, 48)  # 32 field channels + 16 appearance channels


def test_neurad_hip_plugin_training_step_on_the_fused_nodes_matches_the_reference_model(ref, monkeypatch):
    """Training mode of the plugin model (a subclass of the reference's NeuRADModel): get_nff_outputs runs the fused
    training orchestration (FusedTrainMixin: edges from kernel to kernel, ProposalRoundFn per round, NffRenderTrainFn for
    field + head + compositing + appearance) and returns what the reference's own get_nff_outputs returns for the same
    parameters and rays -- and the reference's get_metrics_dict consumes it.  The device is stood in for by the oracle."""
    import nerfstudio.model_components.renderers as ref_renderers
    import nerfstudio.models.neurad as ref_neurad
    from nerfstudio.cameras.rays import RayBundle as RefRayBundle
    from nerfstudio.data.scene_box import SceneBox

    monkeypatch.setattr(ref_neurad, "VGGPerceptualLossPix2Pix", torch.nn.Identity)
    from copy import deepcopy

    from neurad_studio_amd.integration.neurad_hip import neurad_hip

    def shrink(c):
        c.field.grid.static.log2_hashmap_size = 10
        c.field.sdf_beta = 3.0
        for pf in (c.sampling.proposal_field_1, c.sampling.proposal_field_2):
            pf.grid.static.log2_hashmap_size = 9
        c.loss.vgg_mult = 0.0
        return c

    kw = dict(scene_box=SceneBox(aabb=torch.tensor([[-100.0] * 3, [100.0] * 3])), num_train_data=2,
              metadata={"duration": 8.0, "sensor_idx_to_name": {0: "cam", 1: "lidar"}, "trajectories": []})
    torch.manual_seed(0)
    hip = shrink(deepcopy(neurad_hip.config.pipeline.model)).setup(**kw).train()
    ref_cfg = shrink(ref_neurad.NeuRADModelConfig(implementation="torch"))
    for c in (ref_cfg.field, ref_cfg.sampling.proposal_field_1, ref_cfg.sampling.proposal_field_2):
        c.grid.actor.use_4d_hashgrid = False
    refm = ref_cfg.setup(**kw).train()
    with torch.no_grad():
        hip.field.hashgrid.static_grid.hash_table.mul_(500.0)
        for p in hip.proposal_fields:
            p.hashgrid.static_grid.hash_table.mul_(2000.0)
    refm.load_state_dict(hip.state_dict())
    for m in (hip, refm):  # deterministic sampling in training mode, as in oracle/make_golden_model.py
        m.sampler.eval(), m.field.eval()
        for p in m.proposal_fields:
            p.eval()
    na = _tiny_nerfacc()
    monkeypatch.setattr(ref_neurad, "nerfacc", na)
    monkeypatch.setattr(ref_renderers, "nerfacc", na)
    monkeypatch.setattr(type(refm), "_render_weights",
                        lambda self, outputs, rs: na.render_weight_from_alpha(
                            outputs[ref_neurad.FieldHeadNames.ALPHA].squeeze(-1))[0])
    dev = OracleOps()
    dev.install(monkeypatch)
    dev.install_training(monkeypatch)
    R, n_cam = 24, 8
    o, d = _rays(R, seed=7)
    g = np.random.default_rng(3)
    is_lidar = torch.arange(R)[:, None] >= n_cam
    did_return = torch.from_numpy(g.random((R, 1)) < 0.7) | ~is_lidar
    dist = T(g.uniform(2.0, 60.0, (R, 1)).astype(np.float32))

    def bundle():
        return RefRayBundle(origins=T(o), directions=T(d), pixel_area=torch.where(is_lidar, 4.5e-6, 2.7e-7), nears=None,
                            fars=None, times=T(g.uniform(0, 8, (R, 1)).astype(np.float32)) * 0 + 3.3,
                            metadata={"sensor_idxs": is_lidar.long(), "is_lidar": is_lidar, "did_return": did_return,
                                      "directions_norm": dist})

    assert hip.fused_training_possible()
    got = hip.get_nff_outputs(bundle(), calc_lidar_losses=True)
    names = [c[0] for c in dev.calls]
    assert names.count("field_fwd_train") == 1 and names.count("sdf_render_fwd") == 1 and names.count("prop_weights_fwd") == 2
    assert "field_fwd" not in names and names.count("pdf_sample") == 2  # no operator-level field call, two resampling rounds
    want = refm.get_nff_outputs(bundle(), calc_lidar_losses=True)
    for k in ("features", "depth", "accumulation", "prop_depth_0", "prop_depth_1"):
        assert got[k].shape == want[k].shape, k
        assert rel_l2(N(got[k]), N(want[k])) < 2e-4, (k, rel_l2(N(got[k]), N(want[k])))
    for i in range(3):
        assert got["weights_list"][i].shape == want["weights_list"][i].shape
        assert rel_l2(N(got["weights_list"][i]), N(want["weights_list"][i])) < 2e-4, i
    for i in range(2):
        assert abs(float(got[f"prop_weights_loss_{i}"]) - float(want[f"prop_weights_loss_{i}"])) <= 2e-4 * float(want[f"prop_weights_loss_{i}"]) + 1e-9
    # what get_metrics_dict forms from the non-nearby weights (models/neurad.py:508): the dense masked tensor has the same sum
    a, b = float((got["non_nearby_weights"] ** 2).sum()), float((want["non_nearby_weights"] ** 2).sum())
    assert abs(a - b) <= 2e-4 * b + 1e-9
    # the reference's own loss functions read the light RaySamples of the fused path (spacing edges as one tensor)
    from neurad_studio_amd.model_components.losses import ray_samples_to_sdist

    for i in range(3):
        ref_sd = torch.cat([want["ray_samples_list"][i].spacing_starts[..., 0],
                            want["ray_samples_list"][i].spacing_ends[..., -1:, 0]], -1)
        assert rel_l2(N(ray_samples_to_sdist(got["ray_samples_list"][i])), N(ref_sd)) < 1e-5, i



def test_plugin_decoder_adapter_is_the_reference_decoder_call_without_a_copy(ref):
    """NeuRADHipModel.decode_features hands the reference's own method an adapter in place of ``rgb_decoder``: it must return
    what the decoder returns for the NCHW view (on CPU decode_rgb runs the torch modules), and reach the pixel-major rows
    without copying them."""
    from nerfstudio.model_components.cnns import BasicBlock as RefBlock

    from neurad_studio_amd.integration.neurad_hip import _NchwDecoderAdapter
    from neurad_studio_amd.model_components.cnns import _fused_decoder_args

    torch.manual_seed(0)
    dec = torch.nn.Sequential(  # the reference's own construction (models/neurad.py:201-216)
        torch.nn.Conv2d(48, 32, kernel_size=1, padding=0), torch.nn.ReLU(inplace=True),
        RefBlock(32, 32, kernel_size=7, padding=3, use_bn=True), RefBlock(32, 32, kernel_size=7, padding=3, use_bn=True),
        torch.nn.ConvTranspose2d(32, 32, kernel_size=3, stride=3),
        RefBlock(32, 32, kernel_size=7, padding=3, use_bn=True), RefBlock(32, 32, kernel_size=7, padding=3, use_bn=True),
        torch.nn.Conv2d(32, 3, kernel_size=1, padding=0), torch.nn.Sigmoid()).eval()
    args = _fused_decoder_args(dec)  # the HIP path recognises the reference's module tree
    assert args is not None and len(args[0]) == 38 and len(args[1]) == 8
    rows = torch.randn(2 * 8 * 8, 48)
    x = rows.view(2, 8, 8, 48).permute(0, 3, 1, 2)  # what decode_features passes (models/neurad.py:362-363)
    assert x.permute(0, 2, 3, 1).reshape(-1, 48).data_ptr() == rows.data_ptr()
    with torch.no_grad():
        assert torch.equal(_NchwDecoderAdapter(dec)(x), dec(x))


def test_neurad_hip_trainer_config_is_the_references_with_the_trainer_substituted(ref, monkeypatch):
    """``ns-train neurad-hip`` instantiates ``config._target`` (scripts/train.py:96-107 -> TrainerConfig.setup): the method's
    config is a HipTrainerConfig carrying EVERY field of the reference's ``neurad`` trainer config (schedules, logging, steps,
    mixed_precision ...) and naming HipTrainer; the pending-scheduler-step bookkeeping of HipTrainer steps the schedulers
    exactly when the scale did not go down."""
    import dataclasses
    import types

    import nerfstudio.configs.method_configs as ref_methods
    from nerfstudio.engine.trainer import Trainer, TrainerConfig

    from neurad_studio_amd.integration.neurad_hip import neurad_hip
    from neurad_studio_amd.integration.trainer import HipTrainer, HipTrainerConfig

    cfg, ref_cfg = neurad_hip.config, ref_methods.method_configs["neurad"]
    assert isinstance(cfg, HipTrainerConfig) and isinstance(cfg, TrainerConfig) and cfg._target is HipTrainer
    assert issubclass(HipTrainer, Trainer) and cfg.deferred_scheduler_step and cfg.read_only_inf_check
    substituted = {"_target", "method_name", "pipeline", "optimizers"}
    for f in dataclasses.fields(ref_cfg):
        if f.name not in substituted:
            assert getattr(cfg, f.name) == getattr(ref_cfg, f.name), f.name
    assert cfg.mixed_precision is True  # configs/method_configs.py:401
    for name, group in ref_cfg.optimizers.items():  # same groups, same hyper-parameters and schedules
        mine = cfg.optimizers[name]
        assert mine["scheduler"] == group["scheduler"]
        for k in ("lr", "eps", "weight_decay", "max_norm"):
            assert getattr(mine["optimizer"], k) == getattr(group["optimizer"], k), (name, k)

    stepped = []
    loop = object.__new__(HipTrainer)
    loop.optimizers = types.SimpleNamespace(scheduler_step_all=stepped.append)
    done = types.SimpleNamespace(synchronize=lambda: None)
    loop._settle_schedulers()                                   # nothing pending
    loop._pending_scheduler_step = (7, torch.tensor(False), done)
    loop._settle_schedulers()
    loop._settle_schedulers()                                   # (settled once)
    loop._pending_scheduler_step = (8, torch.tensor(True), done)  # the scale went down in iteration 8: no scheduler step
    loop._settle_schedulers()
    assert stepped == [7] and loop._pending_scheduler_step is None
