"""oracle/torch_cpu_port.py (the torch-ops CPU baseline bench.py times on the GPU box's host) against the numpy oracle on a
small case: same field, same rays -> the same rendered features / depth / accumulation to fp32 rounding."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def test_torch_port_matches_the_numpy_oracle():
    import neurad_oracle as O
    import torch_cpu_port as P

    rng = np.random.default_rng(0)
    L, F, lg, H = 8, 4, 12, 32
    grid = O.GridParams((rng.standard_normal((L << lg, F)) * 0.5).astype(np.float32), L, 16, 512, lg)
    geo_w = [(rng.standard_normal((H, L * F)) * 0.3).astype(np.float32), (rng.standard_normal((33, H)) * 0.3).astype(np.float32)]
    feat_w = [(rng.standard_normal((H, 48)) * 0.3).astype(np.float32), (rng.standard_normal((H, H)) * 0.3).astype(np.float32),
              (rng.standard_normal((32, H)) * 0.3).astype(np.float32)]
    for use_sdf in (True, False):
        p = O.FieldParams(grid, 50.0, geo_w, [np.zeros(w.shape[0], np.float32) + 0.01 for w in geo_w], feat_w,
                          [np.zeros(w.shape[0], np.float32) - 0.02 for w in feat_w], beta=3.0, use_sdf=use_sdf)
        R, S = 96, 24
        o = (rng.standard_normal((R, 3)) * 20).astype(np.float32)
        d = rng.standard_normal((R, 3)).astype(np.float32)
        d /= np.linalg.norm(d, axis=-1, keepdims=True)
        area = np.full(R, 3e-7, np.float32)
        edges = np.sort(rng.uniform(0.1, 150.0, (R, S + 1)).astype(np.float32), -1)
        want = O.render_rays(p, o, d, area, edges[:, :-1], edges[:, 1:])
        got = P.render_rays(p, o, d, area, edges[:, :-1], edges[:, 1:])
        for k in ("features", "depth", "accumulation"):
            a, b = got[k].numpy().astype(np.float64), want[k].astype(np.float64)
            assert np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30) < 2e-6, (use_sdf, k)
