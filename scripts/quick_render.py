"""one line per workload: fused render time (HIP events) with the nrhip_ray_order processing order; NEURAD_HIP_LIB selects
an experiment build (scripts/build_variant.sh)"""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "scripts")
import torch
from bench_render_variants import CFGS, mk, rays, timeit
from neurad_studio_amd import ops
out = [os.path.basename(os.environ.get("NEURAD_HIP_LIB", "default"))]
for name in ("config1_16x2_T19_H64_4096x128", "neurad_8x4_T22_H32_4096x128"):
    L, F, lg, H, mn, mx, R, S = CFGS[name]
    fs = mk(L, F, lg, H, mn, mx)
    g = torch.Generator(device="cuda"); g.manual_seed(1234)
    o, d = rays(R, "random", g)
    area = torch.full((R,), 2.43e-6, device="cuda")
    eu = ops.power_sampler(None, torch.full((R,), 20000.0, device="cuda"), S, last_edge=20000.0)[1]
    od = ops.ray_order(o, d, 100.0)
    t = [timeit(lambda: ops.render_fwd(fs, o, d, area, eu[:, :-1], eu[:, 1:], order=od), iters=50) for _ in range(3)]
    out.append(f"{name.split('_')[0]}: {min(t):.1f} us (runs {t})")
print(" | ".join(out))
