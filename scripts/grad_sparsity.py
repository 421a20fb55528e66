"""How sparse are the gradients that reach the table-gradient kernels?  One c1 and one c3 training step of bench.py with
the backward ops wrapped: fraction of samples whose incoming gradient is exactly zero (their records could be skipped
exactly) and fraction below 2^-40 of the largest (below the fixed-point quantum of the reduce pass).
  python scripts/grad_sparsity.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from neurad_studio_amd import ops

dev = torch.device("cuda:0")
seen = []


def wrap(name, gpos, per_sample):
    orig = getattr(ops, name)

    def f(*a, **k):
        g = a[gpos]
        gs = per_sample(g)
        mx = float(gs.max())
        seen.append((name, tuple(g.shape), float((gs == 0).float().mean()), float((gs < mx * 2.0**-40).float().mean()),
                     float((gs < mx * 1e-6).float().mean())))
        return orig(*a, **k)

    setattr(ops, name, f)


wrap("proposal_density_bwd", 7, lambda g: g.abs().reshape(-1))
wrap("encode_bwd", 7, lambda g: g.abs().reshape(-1, 32).amax(-1))
for label, fn in (("c1 train", lambda: bench.train_section(dev, 0, 1, 2, 1)), ("c3 train_full", lambda: bench.train_full_section(dev, 0, 1, 2, 1))):
    seen.clear()
    fn()
    print(label)
    for s in seen[-3:]:
        print("  %-22s grad %-16s exactly zero %.3f   < 2^-40 max %.3f   < 1e-6 max %.3f" % s)
