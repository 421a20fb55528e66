"""HashGridAdam (csrc/adam.hip) against torch.optim.Adam / AdamW: same trajectory, interchangeable state_dict, untouched
rows stay bit-identical."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("wd", [0.0, 1e-2])
def test_hashgrid_adam_matches_torch_adam(wd):
    from neurad_studio_amd.optim import HashGridAdam

    torch.manual_seed(0)
    n_rows, F = 40003, 4  # odd size: exercises the tail
    p0 = (torch.rand(n_rows, F, device="cuda") * 2 - 1) * 1e-3
    a = torch.nn.Parameter(p0.clone())
    b = torch.nn.Parameter(p0.clone())
    ours = HashGridAdam([a], lr=1e-2, eps=1e-15, weight_decay=wd)
    ref = (torch.optim.AdamW([b], lr=1e-2, eps=1e-15, weight_decay=wd) if wd else torch.optim.Adam([b], lr=1e-2, eps=1e-15))
    touched_ever = torch.zeros(n_rows, dtype=torch.bool, device="cuda")
    for step in range(12):
        rows = torch.randint(0, n_rows // 2, (3000,), device="cuda")  # the upper half is never addressed
        g = torch.zeros_like(p0)
        g[rows] = torch.randn(3000, F, device="cuda") * (10.0 ** (step % 4 - 2))
        touched_ever[rows] = True
        a.grad, b.grad = g.clone(), g.clone()
        ours.step()
        ref.step()
        assert torch.allclose(a, b, rtol=2e-6, atol=3e-8), step
    if wd == 0.0:
        assert torch.equal(a[~touched_ever], p0[~touched_ever])  # exact no-op on rows that never saw a gradient
    sa, sb = ours.state[a], ref.state[b]
    assert torch.allclose(sa["exp_avg"], sb["exp_avg"], rtol=1e-6, atol=1e-12)
    assert torch.allclose(sa["exp_avg_sq"], sb["exp_avg_sq"], rtol=1e-6, atol=1e-20)
    assert float(sa["step"]) == float(sb["step"]) == 12
    # state_dict interchange: torch's state continues in ours and the other way round
    c = torch.nn.Parameter(b.detach().clone())
    cont = HashGridAdam([c], lr=1e-2, eps=1e-15, weight_decay=wd)
    import copy

    cont.load_state_dict(copy.deepcopy(ref.state_dict()))  # (a live state_dict shares its step scalar with its owner)
    g = torch.randn_like(p0)
    b.grad, c.grad = g.clone(), g.clone()
    ref.step(), cont.step()
    assert torch.allclose(c, b, rtol=2e-6, atol=3e-8)
    d = torch.nn.Parameter(a.detach().clone())
    back = (torch.optim.AdamW([d], lr=1e-2, eps=1e-15, weight_decay=wd) if wd else torch.optim.Adam([d], lr=1e-2, eps=1e-15))
    back.load_state_dict(copy.deepcopy(ours.state_dict()))
    a.grad, d.grad = g.clone(), g.clone()
    ours.step(), back.step()
    assert torch.allclose(a, d, rtol=2e-6, atol=3e-8)


def test_hashgrid_adam_skips_parameters_without_gradient_and_scales():
    from neurad_studio_amd.optim import HashGridAdam

    a = torch.nn.Parameter(torch.ones(1024, 2, device="cuda"))
    b = torch.nn.Parameter(torch.ones(1024, 2, device="cuda"))
    opt = HashGridAdam([a, b], lr=1e-1)
    a.grad = torch.full_like(a, 8.0)
    opt.step(grad_scale=1.0 / 8.0)
    assert b.grad is None and not opt.state[b] and torch.equal(b, torch.ones_like(b))
    assert torch.allclose(a, torch.full_like(a, 0.9), atol=1e-6)  # first Adam step moves by lr * sign(g)
    assert torch.allclose(opt.state[a]["exp_avg"], torch.full_like(a, 0.1), atol=1e-7)  # (1 - b1) * unscaled g = 0.1


def test_fp16_storage_table_trains_through_the_fused_field_and_master_weights():
    """BASELINE config 5's fp16 hash table under autograd: the fused training forward reads the half table, the table
    gradient is formed in fp32 and arrives in the table's dtype, HashGridAdam updates an fp32 master copy."""
    import synth
    from neurad_studio_amd.cameras.rays import RayBundle
    from neurad_studio_amd.field_components.field_heads import FieldHeadNames
    from neurad_studio_amd.fields.neurad_field import NeuRADField, NeuRADFieldConfig
    from neurad_studio_amd.model_components.ray_samplers import PowerSampler
    from neurad_studio_amd.optim import HashGridAdam

    def make(dtype):
        torch.manual_seed(0)
        cfg = NeuRADFieldConfig()
        cfg.grid.static.log2_hashmap_size = 12
        f = NeuRADField(cfg, actors=None, static_scale=100.0).cuda().train()
        with torch.no_grad():
            f.hashgrid.static_grid.hash_table.mul_(500.0)
        f.hashgrid.static_grid.hash_table.data = f.hashgrid.static_grid.hash_table.data.half().to(dtype)  # same rounded values
        return f

    R, S = 2048, 32
    o, d, area, _ = synth.rays(R, 5)
    dev = lambda a: torch.from_numpy(a).cuda()  # noqa: E731
    rb = RayBundle(origins=dev(o), directions=dev(d), pixel_area=dev(area)[:, None], nears=torch.zeros(R, 1, device="cuda"),
                   fars=torch.full((R, 1), 200.0, device="cuda"))
    rs = PowerSampler(num_samples=S, lambda_=-1.0, scaling=0.1).cuda().eval()(rb)
    outs = {}
    for dtype in (torch.float32, torch.float16):
        f = make(dtype)
        out = f(rs)
        (out[FieldHeadNames.FEATURE].square().mean() + out[FieldHeadNames.ALPHA].mean()).backward()
        g = f.hashgrid.static_grid.hash_table.grad
        assert g.dtype == dtype
        outs[dtype] = (out[FieldHeadNames.FEATURE].detach(), g.float(), f)
    assert torch.equal(outs[torch.float16][0], outs[torch.float32][0])  # same table values -> same forward, bit for bit
    g32, g16 = outs[torch.float32][1], outs[torch.float16][1]
    assert float((g16 - g32).norm() / g32.norm()) < 2e-3  # fp16 rounding of the gradient only
    f = outs[torch.float16][2]
    table = f.hashgrid.static_grid.hash_table
    before = table.detach().clone()
    opt = HashGridAdam([table], lr=1e-2)
    opt.step()
    st = opt.state[table]
    assert st["master"].dtype == torch.float32 and table.dtype == torch.float16
    assert torch.equal(table, st["master"].half()) and float((table.float() - before.float()).abs().max()) > 1e-3


def test_sharded_table_adam_single_process_equals_torch_adam():
    """parallel/sharded_adam.py with one rank: the HIP kernel on the (whole-table) shard == torch.optim.Adam, and its
    state_dict loads into torch.optim.Adam (the N > 1 exchange around it is covered by the gloo world-2 test)"""
    from neurad_studio_amd.parallel.sharded_adam import ShardedTableAdam

    torch.manual_seed(4)
    t = torch.nn.Parameter(torch.randn((1 << 16, 4), device="cuda") * 0.1)
    r = torch.nn.Parameter(t.detach().clone())
    opt, ref = ShardedTableAdam([t], lr=1e-2, eps=1e-15), torch.optim.Adam([r], lr=1e-2, eps=1e-15)
    for it in range(4):
        g = torch.randn_like(t)
        g[torch.rand((t.shape[0],), device="cuda") < 0.7] = 0.0  # untouched rows
        t.grad, r.grad = g.clone(), g.clone()
        assert opt.step() == 0  # nothing on the wire with one rank
        ref.step()
    assert torch.allclose(t, r, rtol=2e-6, atol=3e-8)
    sd = opt.state_dict()
    assert torch.allclose(sd["state"][0]["exp_avg_sq"], ref.state_dict()["state"][0]["exp_avg_sq"], rtol=1e-5, atol=1e-12)
    other = torch.optim.Adam([torch.nn.Parameter(t.detach().clone())], lr=1e-2, eps=1e-15)
    other.load_state_dict(sd)
    assert float(other.state_dict()["state"][0]["step"]) == 4.0
