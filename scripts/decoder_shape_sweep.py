"""ad-hoc: the HIP decoder against the torch modules (fp32) on odd shapes -- forward, input gradient, a few parameter gradients"""
import copy
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurad_studio_amd.model_components.cnns import decode_rgb, make_rgb_decoder  # noqa: E402


def rel(a, b):
    return float((a - b).norm() / (b.norm() + 1e-20))


torch.manual_seed(0)
for cin, B, ph, pw in ((48, 1, 16, 16), (48, 7, 24, 40), (32, 3, 32, 32), (48, 2, 5, 9), (64, 2, 33, 31), (48, 41, 32, 32)):
    dec = make_rgb_decoder(cin, 32, 3).cuda().train()
    f = torch.randn((B * ph * pw, cin), device="cuda")
    img = torch.rand((B, 3 * ph, 3 * pw, 3), device="cuda")
    res = []
    for mode in ("hip", "fp32", "autocast"):
        d = copy.deepcopy(dec)
        x = f.clone().requires_grad_()
        if mode == "autocast":
            with torch.autocast("cuda", dtype=torch.float16):
                rgb = decode_rgb(d, x, (ph, pw), fused=False).float()
        else:
            rgb = decode_rgb(d, x, (ph, pw), fused=mode == "hip")
        (torch.nn.functional.mse_loss(rgb, img) * 1024.0).backward()  # (a loss scale, as the reference's trainer applies one)
        res.append((rgb.detach(), x.grad, {n: p.grad for n, p in d.named_parameters()}))
    (r1, g1, p1), (r0, g0, p0), (ra, ga, pa) = res
    skip = ("main_branch.0.bias", "main_branch.3.bias")
    worst = max(rel(p1[n], p0[n]) for n in p0 if not n.endswith(skip))
    worst_a = max(rel(pa[n], p0[n]) for n in p0 if not n.endswith(skip))
    print(f"cin {cin} B {B} patch {ph}x{pw}: rgb max err {float((r1 - r0).abs().max()):.2e} (autocast {float((ra - r0).abs().max()):.2e})  "
          f"d features {rel(g1, g0):.2e} ({rel(ga, g0):.2e})  worst parameter gradient {worst:.2e} ({worst_a:.2e})", flush=True)
    assert float((r1 - r0).abs().max()) < max(5e-3, 1.5 * float((ra - r0).abs().max()))
    assert rel(g1, g0) < max(2e-2, 1.5 * rel(ga, g0)) and worst < max(3e-2, 1.5 * worst_a)
print("sweep ok")
