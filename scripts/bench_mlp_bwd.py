"""nrhip_mlp_bwd on 524 288 samples for NeuRAD's two MLP shapes (A/B: NRHIP_MLP_SPLIT_WGRAD=1, NRHIP_MLP_GENERIC=1)"""
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0,'tests')
import torch, numpy as np, synth
from neurad_studio_amd import ops
N=524288
for dims in [(32,2,64,33),(48,3,64,32)]:
    i,n,w,o=dims; dd=[i]+[w]*(n-1)+[o]
    ws=[torch.randn(dd[k+1],dd[k],device='cuda')*0.1 for k in range(n)]; bs=[torch.zeros(dd[k+1],device='cuda') for k in range(n)]
    x=torch.randn(N,i,device='cuda'); go=torch.randn(N,o,device='cuda')
    y,h=ops.mlp_fwd(x,ws,bs,save_hidden=True)
    for _ in range(3): ops.mlp_bwd(x,h,go,ws,bs)
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(10): ops.mlp_bwd(x,h,go,ws,bs)
    torch.cuda.synchronize(); print(dims, f"mlp_bwd (data + weight gradients) {(time.perf_counter()-t0)/10*1e3:.3f} ms",
          "split wgrad" if os.environ.get("NRHIP_MLP_SPLIT_WGRAD") else "fused wgrad where it pays")
