import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
import numpy as np, torch
import synth
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
for name,(L,F,lg,H,mn,mx,R,S) in {'C2':(16,2,19,64,16,1024,4096,128),'neurad':(8,4,22,32,32,8192,16384,32)}.items():
    fs = mk(L,F,lg,H,mn,mx)
    o,d,area,_ = synth.rays(R,1)
    eu = ops.power_sampler(None, torch.full((R,), 20000.0, device='cuda'), S)[1].cpu().numpy()
    do,dd,da,ed = dev(o),dev(d),dev(area),dev(eu)
    for fn,label in ((lambda: ops.render_fwd(fs,do,dd,da,ed[:,:-1],ed[:,1:]),'render_fused'),
                     (lambda: ops.field_fwd(fs,do,dd,da,ed[:,:-1],ed[:,1:]),'field_fused'),
                     (lambda: ops.encode_fwd(fs.grid,fs.table,100.0,do,dd,da,ed[:,:-1],ed[:,1:]),'encode_only')):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0,e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)/20
        print(f"{name} {label}: {ms*1e3:.1f} us  -> {R*S/ms/1e3:.3e} samples/s  ({R*S*1033/ms/1e6:.1f} GB/s algorithmic)")
