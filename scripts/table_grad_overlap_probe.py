"""Do the field's and a proposal network's table gradients overlap when they are issued on two HIP streams?  (In the c3
step they are independent branches of the backward: `emit` waits on scattered write requests, `reduce` on the LDS atomic
unit -- two kernels of different chunks could fill each other's stalls.)  c3 shapes, synthetic gradients."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurad_studio_amd import ops  # noqa: E402

R = 57344
dev = "cuda"
g = torch.Generator(device=dev)
g.manual_seed(0)
o = torch.randn((R, 3), device=dev, generator=g) * 5
d = torch.nn.functional.normalize(torch.randn((R, 3), device=dev, generator=g), dim=-1)
area = torch.full((R,), 2.43e-6, device=dev)


def edges(S):
    eu = ops.power_sampler(None, torch.full((R,), 20000.0, device=dev), S)[1]
    return eu[:, :-1].contiguous(), eu[:, 1:].contiguous()


fs = ops.GridSpec(8, 4, 22, 32, 8192)
st_f, en_f = edges(32)
go_f = torch.randn((R * 32, 32), device=dev, generator=g)
pspec = ops.GridSpec(6, 1, 20, 128, 4096)
ptab = (torch.rand(6 * 2**20, 1, device=dev) * 2 - 1) * 0.1
ps = ops.ProposalSpec(pspec, ptab, 100.0, torch.randn(1, 6, device=dev))
st_p, en_p = edges(128)
dens = ops.proposal_density_fwd(ps, o, d, area, st_p, en_p)
gd = torch.randn((R, 128), device=dev, generator=g)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def field(stream=None):
    with torch.cuda.stream(stream or torch.cuda.current_stream()):
        return ops.encode_bwd(fs, 100.0, o, d, area, st_f, en_f, go_f)


def prop(stream=None):
    with torch.cuda.stream(stream or torch.cuda.current_stream()):
        return ops.proposal_density_bwd(ps, o, d, area, st_p, en_p, dens, gd)


def timeit(fns, n=10):
    for f in fns:
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        for f in fns:
            f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


a, b = timeit([field]), timeit([prop])
both1 = timeit([field, prop])
both2 = timeit([lambda: field(sa), lambda: prop(sb)])
pp = timeit([lambda: prop(sa), lambda: prop(sb)])
print(f"field table gradient alone {a:.0f} us | proposal alone {b:.0f} us | back to back on one stream {both1:.0f} us | "
      f"on two streams {both2:.0f} us | two proposal gradients on two streams {pp:.0f} us (2 x alone = {2 * b:.0f})")
