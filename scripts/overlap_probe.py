"""Do a gather-bound kernel (encode_fwd) and an MFMA/stream-bound kernel (chained MLP forward) overlap when they are
issued on two HIP streams?  (Design question for an un-fused, level-partitioned encode feeding the MLP kernel.)"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from neurad_studio_amd import ops
R, S = 4096, 128
N = R * S
spec = ops.GridSpec(16, 2, 19, 16, 1024)
table = (torch.rand(16 * 2**19, 2, device="cuda") * 2 - 1) * 1e-2
g = torch.Generator(device="cuda"); g.manual_seed(0)
o = torch.randn((R, 3), device="cuda", generator=g) * 5
d = torch.randn((R, 3), device="cuda", generator=g); d = d / d.norm(dim=-1, keepdim=True)
area = torch.full((R,), 2.43e-6, device="cuda")
sp, eu = ops.power_sampler(None, torch.full((R,), 20000.0, device="cuda"), S)
st, en = eu[:, :-1].contiguous(), eu[:, 1:].contiguous()
ws = [torch.randn(64, 48, device="cuda") * 0.1, torch.randn(64, 64, device="cuda") * 0.1, torch.randn(32, 64, device="cuda") * 0.1]
bs = [torch.zeros(64, device="cuda"), torch.zeros(64, device="cuda"), torch.zeros(32, device="cuda")]
x = torch.randn(N, 48, device="cuda")
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
def enc():
    with torch.cuda.stream(sa):
        return ops.encode_fwd(spec, table, 100.0, o, d, area, st, en)
def mlp():
    with torch.cuda.stream(sb):
        return ops.mlp_fwd(x, ws, bs)
def timeit(fns, n=20):
    for f in fns: f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        for f in fns: f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6
ta, tb, tab = timeit([enc]), timeit([mlp]), timeit([enc, mlp])
print(f"encode_fwd alone {ta:.0f} us, mlp_fwd alone {tb:.0f} us, both on two streams {tab:.0f} us (sum {ta+tb:.0f}, max {max(ta,tb):.0f})")
def mlp2():
    with torch.cuda.stream(sa):
        return ops.mlp_fwd(x, ws, bs)
print(f"two mlp_fwd on two streams {timeit([mlp, mlp2]):.0f} us; two encode_fwd {timeit([enc, lambda: ops.encode_fwd(spec, table, 100.0, o, d, area, st, en)]):.0f} us")
