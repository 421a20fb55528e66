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


@pytest.mark.parametrize("scaler_cls", ["torch", "table"])
@pytest.mark.parametrize("table_dtype", [torch.float32, torch.float16], ids=["fp32", "fp16-storage"])
def test_hashgrid_adam_speaks_the_gradscaler_protocol(table_dtype, scaler_cls):
    """engine/trainer.py:550-576: every optimizer is stepped through ``grad_scaler.step``.  HashGridAdam declares
    ``_step_supports_amp_scaling`` and consumes ``grad_scale`` / ``found_inf`` on the device (csrc/adam.hip,
    nrhip_adam_step_many_dev): scaled gradients give torch.optim.Adam's trajectory on the unscaled ones, an inf skips the
    step (parameters, moments AND step count untouched) and halves the scale, fp16 gradients are accepted."""
    from neurad_studio_amd.optim import HashGridAdam

    torch.manual_seed(1)
    n_rows, F = 30001, 2
    p0 = ((torch.rand(n_rows, F, device="cuda") * 2 - 1) * 1e-1).half().float()
    a = torch.nn.Parameter(p0.clone().to(table_dtype))
    b = torch.nn.Parameter(p0.clone())
    ours = HashGridAdam([a], lr=1e-2, eps=1e-15)
    ref = torch.optim.Adam([b], lr=1e-2, eps=1e-15)
    from neurad_studio_amd.optim import TableGradScaler

    # (TableGradScaler: the same protocol with the read-only inf check, what HipTrainer installs)
    scaler = (TableGradScaler if scaler_cls == "table" else torch.amp.GradScaler)("cuda", init_scale=1024.0, growth_interval=1000)
    done = 0
    for step in range(8):
        g = torch.zeros_like(p0)
        rows = torch.randint(0, n_rows // 2, (2000,), device="cuda")
        g[rows] = (torch.randn(2000, F, device="cuda") * 0.1).half().float()  # (exactly representable once scaled by 2^k)
        poison = step == 3
        scaler.scale(torch.zeros((), device="cuda"))  # (what grad_scaler.scale(loss) does first: the lazy scale tensor)
        scale = scaler.get_scale()
        a.grad = (g * scale).to(table_dtype)
        if poison:
            a.grad[5, 0] = float("inf")
        before = (a.detach().clone(), {k: v.clone() for k, v in ours.state[a].items()} if ours.state[a] else None)
        scaler.step(ours)
        scaler.update()
        if poison:
            assert torch.equal(a.detach(), before[0])
            for k, v in before[1].items():
                assert torch.equal(ours.state[a][k], v), k
            assert scaler.get_scale() == scale * 0.5
            continue
        b.grad = g.clone()
        ref.step()
        done += 1
        if table_dtype == torch.float32:
            assert torch.allclose(a, b, rtol=2e-6, atol=3e-8), step
        else:
            assert torch.allclose(ours.state[a]["master"], b, rtol=2e-6, atol=3e-8), step
            assert torch.equal(a.detach(), ours.state[a]["master"].half())
    st = ours.state[a]
    assert st["step"].is_cuda and float(st["step"]) == done == 7
    assert torch.allclose(st["exp_avg"], ref.state[b]["exp_avg"], rtol=1e-6, atol=1e-12)


def test_hashgrid_adam_capturable_replays_in_a_hip_graph():
    """capturable=True: step counts and the learning rate behind device pointers -- a captured step replays with no new
    kernel arguments (csrc/adam.hip) and follows torch.optim.Adam with the scheduler's learning rates"""
    from neurad_studio_amd.optim import HashGridAdam

    torch.manual_seed(2)
    p0 = torch.randn(5000, 4, device="cuda") * 0.1
    a, b = torch.nn.Parameter(p0.clone()), torch.nn.Parameter(p0.clone())
    lr = torch.tensor(1e-2, device="cuda")
    ours = HashGridAdam([a], lr=lr, eps=1e-15, capturable=True)
    ref = torch.optim.Adam([b], lr=1e-2, eps=1e-15)
    g_static = torch.zeros_like(p0)
    a.grad = g_static
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        ours.step()  # warm-up outside the capture: state tensors exist, one step taken
    torch.cuda.current_stream().wait_stream(s)
    b.grad = g_static.clone()
    ref.step()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        ours.step()
    for k in range(5):
        g = torch.randn_like(p0)
        g_static.copy_(g)
        new_lr = 1e-2 * 0.9 ** k
        lr.fill_(new_lr)
        graph.replay()
        ref.param_groups[0]["lr"] = new_lr
        b.grad = g.clone()
        ref.step()
        assert torch.allclose(a, b, rtol=3e-6, atol=3e-8), k
    assert float(ours.state[a]["step"]) == 6.0


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16], ids=["fp32", "fp16"])
def test_nonfinite_check_is_torchs_found_inf(dtype):
    """ops.nonfinite_check (nrhip_nonfinite_check_many) against torch._amp_foreach_non_finite_check_and_unscale_ at a scale of
    1: the flag for +inf / -inf / NaN at the first element, the last, the scalar tail and in the middle of a 16-byte group;
    the largest finite values, denormals and -0 do not raise it; 30 tensors (two launches); the tensors are not written."""
    from neurad_studio_amd import ops

    torch.manual_seed(0)
    sizes = [1, 3, 4, 5, 7, 8, 9, 1023, 4096, 65537, (1 << 20) + 3] + [257 + 13 * k for k in range(19)]
    big = torch.finfo(dtype).max
    tiny = torch.finfo(dtype).smallest_normal / 4  # a denormal
    base = []
    for n in sizes:
        t = torch.randn(n, device="cuda").to(dtype)
        t[0], t[-1] = big, -big
        if n > 4:
            t[1], t[2], t[3] = tiny, -0.0, -tiny
        base.append(t)
    one = torch.ones((), device="cuda")

    def both(tensors):
        want = torch.zeros((), device="cuda")
        torch._amp_foreach_non_finite_check_and_unscale_([t.clone() for t in tensors], want, one)
        got = torch.zeros((), device="cuda")
        keep = [t.clone() for t in tensors]
        ops.nonfinite_check(tensors, got)
        for a, b in zip(tensors, keep):
            assert torch.equal(a.view(torch.int16 if dtype == torch.float16 else torch.int32),
                               b.view(torch.int16 if dtype == torch.float16 else torch.int32))
        return float(got), float(want)

    assert both(base) == (0.0, 0.0)
    assert both([]) == (0.0, 0.0)
    for which, where in ((0, 0), (3, 4), (10, (1 << 20) + 2), (10, 1 << 19), (10, (1 << 19) + 1), (29, 100), (7, 1022), (9, 65536),
                         (9, 65532), (24, 0), (23, 5)):
        for bad in (float("inf"), float("-inf"), float("nan")):
            tensors = [t.clone() for t in base]
            tensors[which][where] = bad
            assert both(tensors) == (1.0, 1.0), (which, where, bad)
    # the flag accumulates: a call over clean tensors leaves a raised flag raised
    flag = torch.ones((), device="cuda")
    ops.nonfinite_check(base, flag)
    assert float(flag) == 1.0
    with pytest.raises(ValueError):
        ops.nonfinite_check([base[10][1:]], torch.zeros((), device="cuda"))  # (a view off the 16-byte grid: the caller's job)
