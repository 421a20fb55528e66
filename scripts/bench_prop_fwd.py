import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from neurad_studio_amd import ops
R, S = 16384, 128
spec = ops.GridSpec(6, 1, 20, 128, 4096)
table = (torch.rand(6 * 2**20, 1, device="cuda") * 2 - 1) * 0.1
ps = ops.ProposalSpec(spec, table, 100.0, torch.randn(1, 6, device="cuda"))
g = torch.Generator(device="cuda"); g.manual_seed(0)
o = torch.randn((R, 3), device="cuda", generator=g) * 5
d = torch.randn((R, 3), device="cuda", generator=g); d = d / d.norm(dim=-1, keepdim=True)
area = torch.full((R,), 2.43e-6, device="cuda")
eu = ops.power_sampler(None, torch.full((R,), 20000.0, device="cuda"), S)[1]
st, en = eu[:, :-1].contiguous(), eu[:, 1:].contiguous()
for save in (False, True):
    f = lambda: ops.proposal_density_fwd(ps, o, d, area, st, en, save_features=save)
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): f()
    torch.cuda.synchronize(); print(f"proposal_density_fwd {R}x{S} save_features={save}: {(time.perf_counter()-t0)/20*1e6:.0f} us")
