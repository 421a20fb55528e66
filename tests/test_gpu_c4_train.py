"""BASELINE config[4]'s training mode on a small scene: dynamic actors, appearance embedding, the main field's static table
and actor grids in fp16 STORAGE -- the fused training nodes with row overrides (nrhip_field_fwd_train_ovr), the multi-grid
lookup of the in-box samples, the proposal rounds with the actor overlay (nrhip_actor_density_splice_*) and HashGridAdam's
in-kernel fp16 path (nrhip_adam_step_many).

What pins it: an fp16 table holds exactly the values of the fp32 table it was rounded from, so a model with fp16 storage
must produce the outputs and gradients of the SAME model whose fp32 tables hold those rounded values -- outputs to fp32
rounding, gradients to the fp16 rounding of the gradient itself.  The fp32-storage model in turn is pinned to the
reference's torch model (3-actor scene) by tests/test_gpu_reference_plugin.py and to the reference goldens by
tests/test_gpu_actors.py."""
import copy

import numpy as np
import pytest
import torch

import synth
from conftest import rel_l2

pytestmark = pytest.mark.gpu


def trajectories():
    ts_all = torch.tensor([0.0, 1.0, 2.0, 3.0, 4.0])
    out = []
    for a, (y0, yaw, dims, ts) in enumerate([(8.0, 0.3, (2.0, 4.5, 1.6), ts_all[:3]), (-6.0, -0.2, (2.1, 4.8, 1.7), ts_all),
                                             (-5.0, 0.1, (1.9, 4.2, 1.5), ts_all[1:])]):
        poses = []
        for t in ts:
            c, s = np.cos(yaw + 0.05 * float(t)), np.sin(yaw + 0.05 * float(t))
            p = torch.eye(4)
            p[:3, :3] = torch.tensor([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])
            p[:3, 3] = torch.tensor([12.0 + 2.0 * float(t) + a, y0, 0.5])
            poses.append(p)
        out.append({"timestamps": ts.clone(), "poses": torch.stack(poses), "dims": torch.tensor(dims),
                    "symmetric": torch.tensor(True), "deformable": torch.tensor(False)})
    return out


def build():
    from neurad_studio_amd.model_components.dynamic_actors import DynamicActors, DynamicActorsConfig
    from neurad_studio_amd.models.neurad import NeuRADHotPath, NeuRADHotPathConfig

    c = NeuRADHotPathConfig()
    c.field.grid.static.log2_hashmap_size = 14
    c.field.grid.actor.log2_hashmap_size = 10
    c.field.sdf_beta = 3.0
    for pf in (c.sampling.proposal_field_1, c.sampling.proposal_field_2):
        pf.grid.static.log2_hashmap_size = 12
        pf.grid.actor.log2_hashmap_size = 9
    torch.manual_seed(3)
    m = NeuRADHotPath(c, static_scale=100.0, num_sensors=3, duration=5.0,
                      actors=DynamicActors(DynamicActorsConfig(), trajectories=trajectories())).cuda()
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
    for k, (name, p) in enumerate(m.named_parameters()):
        if name.endswith("hash_table"):
            p.data = T(synth.hash_table(p.shape[0], p.shape[1], seed=100 + k, scale=1.0 if p.shape[1] == 4 else 2.5))
    with torch.no_grad():
        m.field.mlp_geo.layers[-1].bias[0] = 1.2  # a translucent static scene: the rays reach the actors' corridor
    return m


def batch(R=256):
    o = synth.normal((R, 3), 5) * np.array([1.5, 1.5, 0.3], np.float32)
    tgt = np.stack([synth.uniform((R,), 10, 24, 8), np.where(np.arange(R) % 2 == 0, 8.0, -5.5) + synth.uniform((R,), -1.5, 1.5, 9),
                    synth.uniform((R,), 0.0, 1.0, 10)], -1).astype(np.float32)
    d = tgt - o
    d = (d / np.linalg.norm(d, axis=-1, keepdims=True)).astype(np.float32)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
    return dict(o=T(o), d=T(d), area=torch.full((R, 1), 2.7e-7, device="cuda"), times=T(synth.uniform((R,), 0.2, 3.8, 9))[:, None],
                sensor=torch.from_numpy(np.arange(R) % 3)[:, None].cuda(), target=T(synth.uniform((R, 48), 0, 1, 12)))


def step(m, b):
    from neurad_studio_amd.cameras.rays import RayBundle
    from neurad_studio_amd.model_components.losses import distortion_loss, zipnerf_interlevel_loss

    m.train()
    m.sampler.eval(), m.field.eval()  # deterministic: no jitter, no random actor flip
    for p in m.proposal_fields:
        p.eval()
    assert m.fused_training_possible()
    m.zero_grad(set_to_none=True)
    rb = RayBundle(origins=b["o"], directions=b["d"], pixel_area=b["area"].clone(), times=b["times"],
                   metadata={"sensor_idxs": b["sensor"]})
    out = m.get_nff_outputs(rb)
    loss = (5.0 * torch.nn.functional.mse_loss(out["features"], b["target"])
            + 0.001 * zipnerf_interlevel_loss(out["weights_list"], out["ray_samples_list"])
            + 0.002 * distortion_loss(out["weights_list"], out["ray_samples_list"]))
    loss.backward()
    return out, loss


def test_fp16_storage_tables_train_like_their_fp32_images():
    from neurad_studio_amd.optim import HashGridAdam

    a = build()
    hg = a.field.hashgrid
    tables = [hg.static_grid, *hg.actor_grids]
    with torch.no_grad():
        for g in tables:  # fp32 tables holding fp16-representable values
            g.hash_table.data = g.hash_table.data.half().float()
    h = copy.deepcopy(a)
    with torch.no_grad():
        for g in [h.field.hashgrid.static_grid, *h.field.hashgrid.actor_grids]:
            g.hash_table.data = g.hash_table.data.half()
    b = batch()
    out_a, loss_a = step(a, b)
    out_h, loss_h = step(h, b)
    N = lambda t: t.detach().float().cpu().numpy()  # noqa: E731
    for k in ("features", "depth", "accumulation", "prop_depth_0", "prop_depth_1"):
        assert rel_l2(N(out_h[k]), N(out_a[k])) < 2e-6, k
    assert abs(float(loss_h) - float(loss_a)) <= 1e-6 * abs(float(loss_a))
    pa, ph = dict(a.named_parameters()), dict(h.named_parameters())
    n_half = n_actor = 0
    for name, p in pa.items():
        if p.grad is None:
            assert ph[name].grad is None, name
            continue
        g = ph[name].grad
        assert g.dtype == ph[name].dtype
        if g.dtype == torch.float16:
            n_half += 1
            n_actor += "actor_grids" in name
            # the fp16 gradient is the rounded fp32 one: |diff| <= half an ulp of fp16, where the value is a normal number
            ref = p.grad
            tol = ref.abs() * 2.0**-11 + 2.0**-24
            assert bool(((g.float() - ref).abs() <= tol * 1.01).all()), name
        else:
            assert rel_l2(N(g), N(p.grad)) < 2e-5, name
    assert n_half >= 2 and n_actor >= 1, "the actor grids of the rays' corridor must have received a gradient"
    # ---- the optimizer: fp32 master copy + fp16 image written by the kernel --------------------------------------------
    tabs_a = [p for n, p in pa.items() if n.endswith("hash_table") and n.startswith("field.") and p.grad is not None]
    tabs_h = [ph[n] for n, p in pa.items() if n.endswith("hash_table") and n.startswith("field.") and p.grad is not None]
    before = [t.detach().clone() for t in tabs_h]
    oa, oh = HashGridAdam(tabs_a, lr=1e-2, eps=1e-15), HashGridAdam(tabs_h, lr=1e-2, eps=1e-15)
    oa.step(), oh.step()
    for ta, th, b0 in zip(tabs_a, tabs_h, before):
        st = oh.state[th]
        assert st["master"].dtype == torch.float32 and st["exp_avg"].dtype == torch.float32
        assert torch.equal(th, st["master"].half())  # the table IS the rounded master copy
        touched = (th != b0).any(-1)
        assert bool(touched.any()) and not bool(touched.all())  # rows without a gradient were left alone (dead-row skip)
        # first Adam step: |update| = lr where the gradient is non-zero, whatever its magnitude.  Where the fp16 gradient is
        # non-zero the two tables therefore move alike; where it UNDERFLOWED (|g| < 2^-25: fp16 has no such number, the
        # reference's GradScaler exists for that) the fp16-storage table stays put
        g16 = th.grad.float()
        moved = g16 != 0
        assert float(((st["master"] - ta).abs() * moved).max()) <= 1e-2 * 1e-3 + 1e-7
        assert torch.equal(st["master"][~moved], b0.float()[~moved])
    # a second step from the fp16 gradients again: state carried, tables stay the image of the master
    step(h, b)
    oh.step()
    for th in tabs_h:
        assert torch.equal(th, oh.state[th]["master"].half()) and int(oh.state[th]["step"]) == 2


def test_adam_step_many_matches_torch_adam():
    from neurad_studio_amd import ops

    g = torch.Generator(device="cuda").manual_seed(5)
    shapes = [(1000, 4), (4096, 1), (513, 2), (7,), (2048, 4)] * 7  # 35 tensors: two launches of <= 24
    ps = [torch.randn(s, device="cuda", generator=g) for s in shapes]
    gs = [torch.randn(s, device="cuda", generator=g) * (torch.rand(s, device="cuda", generator=g) > 0.3) for s in shapes]
    ref = [torch.nn.Parameter(p.clone()) for p in ps]
    opt = torch.optim.Adam(ref, lr=1e-2, eps=1e-15)
    ms, vs = [torch.zeros_like(p) for p in ps], [torch.zeros_like(p) for p in ps]
    for it in range(1, 4):
        for r, gr in zip(ref, gs):
            r.grad = gr.clone() * it
        opt.step()
        ops.adam_step_many([(p, gr * it, m, v, it, None) for p, gr, m, v in zip(ps, gs, ms, vs)], 1e-2)
    for p, r in zip(ps, ref):
        assert torch.allclose(p, r.detach(), rtol=2e-6, atol=2e-7)
