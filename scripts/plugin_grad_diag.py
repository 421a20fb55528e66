"""Per-loss-term comparison of the plugin's gradients (MI355X) with the reference torch model's (CPU): which term of
get_loss_dict carries the difference.  Needs oracle/_ref (or /root/reference).  python scripts/plugin_grad_diag.py [actors]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import torch

import test_gpu_reference_plugin as t

t.ref_import.install()
os.environ["NERFSTUDIO_METHOD_CONFIGS"] = "neurad-hip=neurad_studio_amd.integration.neurad_hip:neurad_hip"
import nerfstudio.models.neurad as ref_neurad

ref_neurad.VGGPerceptualLossPix2Pix = torch.nn.Identity
wa = "actors" in sys.argv
fused_dec = "fdec" in sys.argv
torch.backends.cudnn.allow_tf32 = False
torch.backends.cuda.matmul.allow_tf32 = False
hip, refm = t._build_pair(ref_neurad, wa, fused_decoder=fused_dec)
b = t._batch(wa)
t._deterministic(hip, True), t._deterministic(refm, True)


def run(m, dev):
    m.zero_grad(set_to_none=True)
    out = m.get_outputs(t._bundle(b, dev), patch_size=(b["patch"], b["patch"]), calc_lidar_losses=True)
    lab = t._labels(b, dev)
    met = m.get_metrics_dict(out, lab)
    return out, m.get_loss_dict(out, lab, met)


go, gl = run(hip, "cuda")
wo, wl = run(refm, "cpu")
names = [n for n, p in hip.named_parameters() if not n.startswith("rgb_decoder")]
hp, rp = dict(hip.named_parameters()), dict(refm.named_parameters())
for k in wl:
    gg = torch.autograd.grad(gl[k], [hp[n] for n in names], retain_graph=True, allow_unused=True)
    wg = torch.autograd.grad(wl[k], [rp[n] for n in names], retain_graph=True, allow_unused=True)
    print(f"== {k}: {float(gl[k]):.6e} vs {float(wl[k]):.6e}")
    for n, a, c in zip(names, gg, wg):
        if c is None or float(c.abs().max()) == 0:
            if a is not None and float(a.abs().max()) > 0:
                print(f"   {n}: reference None/0, plugin |g|={float(a.norm()):.3e}")
            continue
        if a is None:
            print(f"   {n}: plugin None, reference |g|={float(c.norm()):.3e}")
            continue
        e = t.rel_l2(t.N(a), t.N(c))
        if e > 1e-5:
            print(f"   {n}: rel-L2 {e:.2e}  (|ref| {float(c.norm()):.3e})")
