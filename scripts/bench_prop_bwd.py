"""proposal-field table gradient (S2 backward) on a NeuRAD-sized proposal batch: binned vs memory-side atomics"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from neurad_studio_amd import ops
R, S = 16384, 128
spec = ops.GridSpec(6, 1, 20, 128, 4096)
table = (torch.rand(6 * 2**20, 1, device="cuda") * 2 - 1) * 0.1
dec = torch.randn(1, 6, device="cuda")
ps = ops.ProposalSpec(spec, table, 100.0, dec)
g = torch.Generator(device="cuda"); g.manual_seed(0)
o = torch.randn((R, 3), device="cuda", generator=g) * 5
d = torch.randn((R, 3), device="cuda", generator=g); d = d / d.norm(dim=-1, keepdim=True)
area = torch.full((R,), 2.43e-6, device="cuda")
sp, eu = ops.power_sampler(None, torch.full((R,), 20000.0, device="cuda"), S)
st, en = eu[:, :-1].contiguous(), eu[:, 1:].contiguous()
dens = ops.proposal_density_fwd(ps, o, d, area, st, en)
gd = torch.randn((R, S), device="cuda", generator=g)
for atomic in (False, True):
    ops._FORCE_ATOMIC_SCATTER = atomic
    for _ in range(2): gt, gdec = ops.proposal_density_bwd(ps, o, d, area, st, en, dens, gd)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): gt, gdec = ops.proposal_density_bwd(ps, o, d, area, st, en, dens, gd)
    torch.cuda.synchronize()
    print(f"proposal_density_bwd {R}x{S}, L=6 T=2^20: {(time.perf_counter()-t0)/5*1e3:.3f} ms  atomic={atomic}  checksum {gt.double().sum().item():.6e} {gdec.double().sum().item():.6e}")
