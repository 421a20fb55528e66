"""Upper bound of what a cache-friendly layout of the COARSE levels could buy the fused render kernel: time it with the
scalings of the first k levels set to 1 (all samples then read the same 8 entries of those levels: perfect hits), k = 0
(the real grid) .. L.  Not a product path; the outputs of the k > 0 runs are meaningless."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from bench_render_variants import CFGS, mk, rays, timeit
from neurad_studio_amd import ops

for name in ("config1_16x2_T19_H64_4096x128", "neurad_8x4_T22_H32_4096x128"):
    L, F, lg, H, mn, mx, R, S = CFGS[name]
    g = torch.Generator(device="cuda"); g.manual_seed(1234)
    o, d = rays(R, "random", g)
    area = torch.full((R,), 2.43e-6, device="cuda")
    eu = ops.power_sampler(None, torch.full((R,), 20000.0, device="cuda"), S, last_edge=20000.0)[1]
    od = ops.ray_order(o, d, 100.0)
    fs = mk(L, F, lg, H, mn, mx)
    real = fs.grid.scalings.clone()
    pts = [(int(s) + 1) ** 3 for s in real.tolist()]
    row = []
    for k in range(0, L + 1):
        fs.grid.scalings = real.clone()
        fs.grid.scalings[:k] = 1.0
        t = min(timeit(lambda: ops.render_fwd(fs, o, d, area, eu[:, :-1], eu[:, 1:], order=od), iters=40) for _ in range(2))
        row.append(f"{k}:{t:.0f}")
    print(name.split("_")[0], "lattice points per level", pts, "T", 1 << lg)
    print("   first k levels made free -> kernel us:", " ".join(row))
