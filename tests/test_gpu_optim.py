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
