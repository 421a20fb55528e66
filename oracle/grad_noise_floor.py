"""Reference-vs-reference noise floor of the composed training step's gradients: the reference's own torch model in fp64
against the same model in fp32, on the batches of tests/test_gpu_reference_plugin.py (build container or GPU box, CPU only),
per loss term of get_loss_dict and per kind of parameter.  The composed-step gradient tolerances of that test are set from
these numbers (3x the floor, at least 1e-4).

What the floor is made of: the LOSS VALUES of the two precisions agree to 1e-7, the gradients of the lidar terms only to
~5e-3 -- a handful of the 2816 x 32 hidden units of mlp_geo sit within 5e-5 of the ReLU kink (the fp32 positions of far
samples move the level-8192 features by that much), flip between the two runs, and each flip switches one sample's
whole contribution on or off.  Terms whose gradient is spread over many samples (rgb, interlevel) agree to 2e-5.
TEST INFRASTRUCTURE.   python oracle/grad_noise_floor.py  ->  profiles/r04_grad_noise_floor.json"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), HERE]
import numpy as np
import torch

import test_gpu_reference_plugin as t

t.ref_import.install()
os.environ["NERFSTUDIO_METHOD_CONFIGS"] = "neurad-hip=neurad_studio_amd.integration.neurad_hip:neurad_hip"
import nerfstudio.models.neurad as ref_neurad

ref_neurad.VGGPerceptualLossPix2Pix = torch.nn.Identity
torch.nn.Module.to = lambda self, *a, **k: self  # (_build_pair moves the plugin to cuda: not here)
out = {}
for wa in (False, True):
    _, m32 = t._build_pair(ref_neurad, wa)
    _, m64 = t._build_pair(ref_neurad, wa)
    m64 = m64.double()
    b = t._batch(wa)
    t._deterministic(m32, True), t._deterministic(m64, True)

    def step(m, dt):
        m.zero_grad(set_to_none=True)
        rb = t._bundle(b, "cpu")
        lab = t._labels(b, "cpu")
        if dt == torch.float64:
            for k in ("origins", "directions", "pixel_area", "times"):
                setattr(rb, k, getattr(rb, k).double())
            rb.metadata["directions_norm"] = rb.metadata["directions_norm"].double()
            lab = {k: (v.double() if v.is_floating_point() else v) for k, v in lab.items()}
        o = m.get_outputs(rb, patch_size=(b["patch"], b["patch"]), calc_lidar_losses=True)
        met = m.get_metrics_dict(o, lab)
        ls = m.get_loss_dict(o, lab, met)
        return ls

    floors = t.per_loss_gradient_errors(m32, step(m32, torch.float32), m64, step(m64, torch.float64))
    out["actors3" if wa else "static"] = floors
    print("actors3" if wa else "static", json.dumps({k: {a: float(f"{b:.2e}") for a, b in v.items()} for k, v in floors.items()}))
json.dump(out, open(os.path.join(ROOT, "profiles", "r04_grad_noise_floor.json"), "w"), indent=1)
