"""Whole NeuRAD hot-path training step at the reference's default sizes (models/neurad.py defaults: static grid L=8,
F=4, T=2^22; proposal grids L=6, F=1, T=2^20; 128+64 proposal samples, 32 field samples; 32-wide MLPs):
get_nff_outputs (training mode, jitter) -> feature/depth stand-ins + the reference's sampler losses (zipnerf interlevel,
distortion, with NeuRAD's multipliers, models/neurad.py:83-85) -> backward -> Adam.
Prints ms/step; run under rocprofv3 --kernel-trace --stats for the per-kernel split.

  python scripts/bench_full_train.py [n_rays] [steps]
"""
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import torch

from neurad_studio_amd.cameras.rays import RayBundle
from neurad_studio_amd.model_components.losses import distortion_loss, zipnerf_interlevel_loss
from neurad_studio_amd.models.neurad import NeuRADHotPath, NeuRADHotPathConfig

R = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda")
torch.manual_seed(0)
cfg = NeuRADHotPathConfig(appearance_dim=0)
m = NeuRADHotPath(cfg, static_scale=100.0).to(dev).train()
with torch.no_grad():
    m.field.hashgrid.static_grid.hash_table.mul_(1000.0)
    for p in m.proposal_fields:
        p.hashgrid.static_grid.hash_table.mul_(2000.0)
params = [p for p in m.parameters() if p.requires_grad]
try:
    opt = torch.optim.Adam(params, lr=1e-3, eps=1e-15, fused=True)
except (RuntimeError, TypeError):
    opt = torch.optim.Adam(params, lr=1e-3, eps=1e-15)
g = torch.Generator(device=dev)
g.manual_seed(1)
o = torch.randn((R, 3), device=dev, generator=g) * 20.0
d = torch.randn((R, 3), device=dev, generator=g)
d = d / d.norm(dim=-1, keepdim=True)
target = torch.rand((R, 32), device=dev, generator=g)


def step():
    rb = RayBundle(origins=o, directions=d, pixel_area=torch.full((R, 1), 2.43e-6, device=dev),
                   nears=torch.zeros((R, 1), device=dev), fars=torch.full((R, 1), 20000.0, device=dev))
    out = m.get_nff_outputs(rb)
    loss = (out["features"] - target).square().mean() + 1e-3 * out["depth"].abs().mean()
    loss = loss + 0.001 * zipnerf_interlevel_loss(out["weights_list"], out["ray_samples_list"])
    loss = loss + 0.002 * distortion_loss(out["weights_list"], out["ray_samples_list"])
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()
    return loss


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(STEPS):
    loss = step()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / STEPS * 1e3
n_field = R * cfg.sampling.num_nerf_samples
n_prop = R * sum(cfg.sampling.num_proposal_samples)
print(f"full train step: {R} rays ({n_prop} proposal + {n_field} field samples): {ms:.3f} ms/step, "
      f"{R / ms * 1e3:.3e} rays/s, loss {loss.item():.4e}")
