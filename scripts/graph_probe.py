"""Does a HIP graph of the config[1] eval step (PowerSampler bins + ray ordering + fused render: three launches) beat the
eager launches?  python scripts/graph_probe.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from neurad_studio_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
fs, origins, dirs, area, fars = bench.make_workload(dev, seed=1234)
S = bench.N_SAMPLES
feats = torch.empty((bench.R_RAYS, 32), device=dev)
depth = torch.empty((bench.R_RAYS, 1), device=dev)
acc = torch.empty((bench.R_RAYS, 1), device=dev)


def step():
    sp, eu = ops.power_sampler(None, fars, S, lam=-1.0, scaling=0.1, last_edge=20000.0)
    order = ops.ray_order(origins, dirs, bench.STATIC_SCALE)
    ops.render_fwd(fs, origins, dirs, area, eu[:, :-1], eu[:, 1:], out=(feats, depth, acc), order=order)


def timeit(fn, n=300):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


eager = timeit(step)
ref = feats.clone()
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        step()
torch.cuda.current_stream().wait_stream(side)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    step()
feats.zero_()
g.replay()
torch.cuda.synchronize()
print("graph output identical:", torch.equal(feats, ref))
print(f"eager {eager:.4f} ms/step   graph {timeit(g.replay):.4f} ms/step   eager again {timeit(step):.4f}")
