"""__graft_entry__.smoke(): one small invocation of the hot path on cuda:0, checked against the oracle."""
from __future__ import annotations

import numpy as np
import torch


def run() -> None:
    import neurad_oracle as O  # checker only (oracle/ is test infrastructure)
    import synth

    from . import ops

    assert torch.cuda.is_available(), "smoke() needs a GPU"
    torch.cuda.set_device(0)
    L, F, lg, H, R, S = 8, 4, 12, 32, 64, 32
    grid = O.GridParams(synth.hash_table(L * 2**lg, F, seed=1, scale=1.0), L, 32, 8192, lg)
    gw, gb, fw, fb = [], [], [], []
    for k, (o, i) in enumerate([(H, 32), (33, H)]):
        w, b = synth.linear(o, i, 10 + k)
        gw.append(w), gb.append(b)
    for k, (o, i) in enumerate([(H, 48), (H, H), (32, H)]):
        w, b = synth.linear(o, i, 20 + k)
        fw.append(w), fb.append(b)
    p = O.FieldParams(grid, 100.0, gw, gb, fw, fb, beta=3.0)
    o, d, area, _ = synth.rays(R, 3)
    _, eu, _ = O.power_sampler(np.zeros(R), np.full(R, 150.0, np.float32), S)
    ref = O.render_rays(p, o, d, area, eu[:, :-1], eu[:, 1:])
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
    fs = ops.FieldSpec(ops.GridSpec(L, F, lg, 32, 8192), dev(grid.table), 100.0, [dev(w) for w in gw],
                       [dev(b) for b in gb], [dev(w) for w in fw], [dev(b) for b in fb], True, abs(p.beta) + p.beta_min)
    edges = dev(eu)
    feats, depth, acc = ops.render_fwd(fs, dev(o), dev(d), dev(area), edges[:, :-1], edges[:, 1:])
    torch.cuda.synchronize()
    err = float(np.linalg.norm(feats.cpu().numpy() - ref["features"]) / np.linalg.norm(ref["features"]))
    assert err < 1e-4, f"smoke: fused render kernel vs oracle rel-L2 {err}"
    print(f"smoke OK: render_fwd {R}x{S} rel-L2 vs oracle = {err:.2e}")
