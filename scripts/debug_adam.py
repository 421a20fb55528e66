import sys, os
sys.path.insert(0, os.getcwd())
import torch
from neurad_studio_amd.optim import HashGridAdam
torch.manual_seed(0)
n_rows, F = 40003, 4
p0 = (torch.rand(n_rows, F, device="cuda") * 2 - 1) * 1e-3
a = torch.nn.Parameter(p0.clone()); b = torch.nn.Parameter(p0.clone())
ours = HashGridAdam([a], lr=1e-2, eps=1e-15); ref = torch.optim.Adam([b], lr=1e-2, eps=1e-15)
for step in range(4):
    rows = torch.randint(0, n_rows // 2, (3000,), device="cuda")
    g = torch.zeros_like(p0); g[rows] = torch.randn(3000, F, device="cuda") * (10.0 ** (step % 4 - 2))
    a.grad, b.grad = g.clone(), g.clone()
    ours.step(); ref.step()
    d = (a - b).abs(); i = int(d.argmax())
    r, c = divmod(i, F)
    print(step, "max abs diff", float(d.max()), "at", r, c, "a", float(a[r, c]), "b", float(b[r, c]), "g", float(g[r, c]),
          "m", float(ours.state[a]["exp_avg"][r, c]), float(ref.state[b]["exp_avg"][r, c]),
          "v", float(ours.state[a]["exp_avg_sq"][r, c]), float(ref.state[b]["exp_avg_sq"][r, c]))
