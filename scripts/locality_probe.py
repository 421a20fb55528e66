"""How much of render_kernel's time is the gather?  Same kernel, same work, different memory behaviour."""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
import numpy as np, torch
import synth
from neurad_studio_amd import ops
dev = lambda a, dt=torch.float32: torch.from_numpy(np.ascontiguousarray(a)).to('cuda', dt)
def lin(o,i,s):
    w,b = synth.linear(o,i,s); return dev(w), dev(b)
def mk(L,F,lg,H,mn,mx, half=False):
    spec = ops.GridSpec(L,F,lg,mn,mx)
    table = (torch.rand(L*2**lg, F, device='cuda')*2-1)*1e-3
    if half: table = table.half()
    gw0,gb0 = lin(H,32,1); gw1,gb1 = lin(33,H,2)
    fw0,fb0 = lin(H,48,3); fw1,fb1 = lin(H,H,4); fw2,fb2 = lin(32,H,5)
    return ops.FieldSpec(spec, table, 100.0, [gw0,gw1],[gb0,gb1],[fw0,fw1,fw2],[fb0,fb1,fb2], True, 20.0001)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0,e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n*1e3
R,S = 4096,128
o,d,area,_ = synth.rays(R,1)
eu = ops.power_sampler(None, torch.full((R,), 20000.0, device='cuda'), S)[1].cpu().numpy()
ed = dev(eu); da = dev(area)
for name, lg, oo, dd, half in [("C2 random rays T=2^19", 19, o, d, False), ("C2 identical rays T=2^19", 19, np.repeat(o[:1],R,0), np.repeat(d[:1],R,0), False),
                         ("C2 random rays T=2^12 (L2 resident)", 12, o, d, False), ("C2 random rays T=2^19 fp16 table", 19, o, d, True),
                         ("C2 random rays T=2^22", 22, o, d, False)]:
    fs = mk(16,2,lg,64,16,1024, half)
    do, ddv = dev(oo), dev(dd)
    us = timeit(lambda: ops.render_fwd(fs,do,ddv,da,ed[:,:-1],ed[:,1:]))
    print(f"{name:42s} {us:8.1f} us")

# every sample of every ray at the same point: each gather instruction touches <= 4 lines (one per level group)
fs = mk(16,2,19,64,16,1024)
do, ddv = dev(np.repeat(o[:1],R,0)), dev(np.repeat(d[:1],R,0))
st = torch.full((R,S), 5.0, device='cuda'); en = torch.full((R,S), 5.01, device='cuda')
us = timeit(lambda: ops.render_fwd(fs,do,ddv,da,st,en))
print(f"{'C2 all samples at ONE point':42s} {us:8.1f} us")
