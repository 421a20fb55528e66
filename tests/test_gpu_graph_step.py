"""A whole training step of the hot path captured in a HIP graph (torch.cuda.graph) and replayed: no device->host read, no
new kernel argument between replays -- step counts, jitter draws and (optionally) the learning rate live on the device.
Replays must walk the trajectory of the eagerly launched step (bench.py's train_full times the replay)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(capturable):
    from test_gpu_dist_rehearsal import _loss, _model, _shard

    from neurad_studio_amd.optim import HashGridAdam

    m = _model()  # (sampler in eval mode: no jitter, so that graph and eager runs see the same samples)
    params = [p for p in m.parameters() if p.requires_grad]
    tables = [p for p in params if p.numel() >= 1 << 14]
    small = [p for p in params if p.numel() < 1 << 14]
    kw = {"capturable": True} if capturable else {}
    opts = [HashGridAdam(tables, lr=1e-2, eps=1e-3, **kw), torch.optim.Adam(small, lr=1e-2, eps=1e-3, fused=True, **kw)]
    from neurad_studio_amd.cameras.rays import RayBundle

    src = _shard(0, 0)  # the batch's tensors live on the device; every step builds its own bundle from them (the model
    state = {}          # scales pixel_area and clamps fars IN the bundle it is given: models/neurad.py:443-449,702-709)

    def step():
        for o in opts:
            o.zero_grad(set_to_none=True)
        rb = RayBundle(origins=src.origins, directions=src.directions, pixel_area=src.pixel_area.clone(),
                       nears=src.nears.clone(), fars=None, times=src.times, metadata=dict(src.metadata))
        loss = _loss(m, rb)
        loss.backward()
        for o in opts:
            o.step()
        state["loss"] = loss

    return m, step, state


def test_captured_training_step_replays_the_eager_trajectory():
    m_e, step_e, st_e = _setup(False)
    for _ in range(6):
        step_e()
    want = {n: p.detach().clone() for n, p in m_e.named_parameters()}
    m_g, step_g, st_g = _setup(True)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):  # warm-up on a side stream (the AccumulateGrad nodes then live where the capture runs)
        step_g()
        step_g()
    torch.cuda.current_stream().wait_stream(s)
    st_g.clear()
    after_warmup = {n: p.detach().clone() for n, p in m_g.named_parameters()}
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        step_g()
    for _ in range(4):
        graph.replay()
    torch.cuda.synchronize()
    assert torch.isfinite(st_g["loss"])
    # a second eager run is the yardstick: the MLP weight gradients are summed with atomics, their order differs from run to
    # run and Adam carries the difference along (two eager runs of six steps agree to ~1e-4, not bit for bit)
    m_e2, step_e2, _ = _setup(False)
    for _ in range(6):
        step_e2()
    for n, p in m_g.named_parameters():
        a, b = p.detach().double(), want[n].double()
        yard = float((dict(m_e2.named_parameters())[n].detach().double() - b).norm() / (b.norm() + 1e-30))
        err = float((a - b).norm() / (b.norm() + 1e-30))
        moved = float((a - after_warmup[n].double()).norm() / (b.norm() + 1e-30))
        assert err < max(5.0 * yard, 2e-5), (n, err, yard)
        assert moved > 20 * err or moved == 0.0, (n, moved, err)  # the replays really trained (4 of the 6 steps)
