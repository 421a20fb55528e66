"""Fused render kernel and table gradient under the bench's spread-out rays vs. a street-like distribution (origins in a
40 m disc 1.5 m above ground, near-horizontal directions, 150 m range): real scenes concentrate the samples in a slab."""
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
import numpy as np, torch, synth
from neurad_studio_amd import ops
dev = lambda a, dt=torch.float32: torch.from_numpy(np.ascontiguousarray(a)).to('cuda', dt)
def lin(o,i,s):
    w,b = synth.linear(o,i,s); return dev(w), dev(b)
def mk(L,F,lg,H,mn,mx):
    spec = ops.GridSpec(L,F,lg,mn,mx)
    table = (torch.rand(L*2**lg, F, device='cuda')*2-1)*1e-3
    gw0,gb0 = lin(H,32,1); gw1,gb1 = lin(33,H,2)
    fw0,fb0 = lin(H,48,3); fw1,fb1 = lin(H,H,4); fw2,fb2 = lin(32,H,5)
    return ops.FieldSpec(spec, table, 100.0, [gw0,gw1],[gb0,gb1],[fw0,fw1,fw2],[fb0,fb1,fb2], True, 20.0001)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n*1e6
R,S = 4096,128
g = torch.Generator(device='cuda'); g.manual_seed(0)
for name,(L,F,lg,H,mn,mx) in {"config 2":(16,2,19,64,16,1024), "NeuRAD default":(8,4,22,32,32,8192)}.items():
    fs = mk(L,F,lg,H,mn,mx)
    for dist in ("bench (spread)", "street (slab, 150 m)"):
        if dist.startswith("bench"):
            o = torch.randn((R,3), device='cuda', generator=g)*5; d = torch.randn((R,3), device='cuda', generator=g); far=20000.0
        else:
            o = torch.randn((R,3), device='cuda', generator=g)*20; o[:,2]=1.5
            d = torch.randn((R,3), device='cuda', generator=g); d[:,2]*=0.1; far=150.0
        d = d/d.norm(dim=-1,keepdim=True)
        area = torch.full((R,),2.43e-6,device='cuda')
        eu = ops.power_sampler(None, torch.full((R,),far,device='cuda'), S)[1]
        st,en = eu[:,:-1],eu[:,1:]
        go = torch.randn((R*S, L*F), device='cuda', generator=g)
        tr = timeit(lambda: ops.render_fwd(fs,o,d,area,st,en))
        tb = timeit(lambda: ops.encode_bwd(fs.grid,100.0,o,d,area,st,en,go))
        ops._FORCE_ATOMIC_SCATTER=True
        ta = timeit(lambda: ops.encode_bwd(fs.grid,100.0,o,d,area,st,en,go), n=5)
        ops._FORCE_ATOMIC_SCATTER=False
        print(f"{name:16s} {dist:22s} render {tr:6.1f} us | table gradient: radix partition {tb:7.1f} us, atomics {ta:8.1f} us")
