import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from neurad_studio_amd import ops
dev = lambda a, dt=torch.float32: torch.from_numpy(np.ascontiguousarray(a)).to('cuda', dt)
for kk in (3.0, 5.0):
    for ax in (0, 1):
        pts = np.zeros((1, 3), np.float32); pts[0, ax] = np.float32(kk) / np.float32(8191)
        spec = ops.GridSpec(1, 1, 14, 8191, 8191)
        gt = ops.hashgrid_bwd(spec, None, dev(pts), dev(np.ones((1, 1), np.float32))).cpu().numpy()[:, 0]
        nz = np.nonzero(gt)[0]
        print("k", kk, "axis", ax, "x", repr(pts[0, ax]), "->", [(int(i), repr(gt[i])) for i in nz])
