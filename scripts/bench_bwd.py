"""micro-benchmark of the hash-table scatter-add backward (encode_bwd) on the bench workload"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from neurad_studio_amd import ops
R, S = 4096, 128
cfg = (16, 2, 19, 16, 1024) if len(sys.argv) < 2 else tuple(int(a) for a in sys.argv[1:6])
spec = ops.GridSpec(*cfg)
g = torch.Generator(device="cuda"); g.manual_seed(0)
o = torch.randn((R, 3), device="cuda", generator=g) * 5
d = torch.randn((R, 3), device="cuda", generator=g); d = d / d.norm(dim=-1, keepdim=True)
area = torch.full((R,), 2.43e-6, device="cuda")
sp, eu = ops.power_sampler(None, torch.full((R,), 20000.0, device="cuda"), S)
go = torch.randn((R * S, cfg[0] * cfg[1]), device="cuda", generator=g)
def run():
    return ops.encode_bwd(spec, 100.0, o, d, area, eu[:, :-1], eu[:, 1:], go)
for _ in range(3): gt = run()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): gt = run()
torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 10 * 1e3
print(f"encode_bwd {cfg}: {ms:.3f} ms  (atomic={os.environ.get('NRHIP_ENCODE_BWD_ATOMIC')}) checksum {gt.double().sum().item():.6e} {gt.double().abs().sum().item():.6e}")
