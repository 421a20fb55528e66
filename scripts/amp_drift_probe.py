"""Which half of mixed precision moves the plugin's K-iteration loop away from the reference's fp32 loop (static scene, full
learning rates, tests/test_gpu_plugin_train_loop.py's harness): autocast alone, GradScaler alone, both, neither.
usage (GPU box): python scripts/amp_drift_probe.py > gpurun_out/amp_drift_probe.txt"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

import ref_import

ref_import.install()
os.environ.setdefault("NERFSTUDIO_METHOD_CONFIGS", "neurad-hip=neurad_studio_amd.integration.neurad_hip:neurad_hip")
import test_gpu_plugin_train_loop as L
import test_gpu_reference_plugin as t
from torch.cuda.amp.grad_scaler import GradScaler

import nerfstudio.models.neurad as ref_neurad

ref_neurad.VGGPerceptualLossPix2Pix = torch.nn.Identity
methods = L._methods()
K = L.K


def run(who, autocast, scaler, device="cuda:0"):
    hip, ref32 = t._build_pair(ref_neurad, False)
    m = hip if who == "hip" else (ref32.to(device) if device != "cpu" else ref32)
    if who != "hip" and device != "cpu":
        m.camera_optimizer.to(device)
    t._deterministic(m, True)
    loop = L._Loop(methods["neurad-hip" if who == "hip" else "neurad"], m, L._Pipeline(m, False, device.split(":")[0], torch.float32),
                   device, True, warmup=False)
    loop.mixed_precision = bool(autocast)                       # what Trainer.train_iteration hands torch.autocast
    loop.grad_scaler = GradScaler(enabled=bool(scaler))         # and what scales the loss
    return loop.run(K)


base = run("ref", False, False, "cpu")
rows = {"hip: fp32 (no autocast, no scaler)": run("hip", False, False), "hip: GradScaler only": run("hip", False, True),
        "hip: autocast only": run("hip", True, False), "hip: autocast + GradScaler": run("hip", True, True),
        "reference on the GPU: fp32": run("ref", False, False), "reference on the GPU: GradScaler only": run("ref", False, True),
        "reference on the GPU: autocast only": run("ref", True, False), "reference on the GPU: autocast + GradScaler": run("ref", True, True)}
terms = ("depth_loss", "rgb_loss", "interlevel_loss", "intensity_loss")
print("# relative error of loss terms against the reference's fp32 CPU loop, iterations 3 / 6 / 9 (static scene, no warm-up)")
for name, losses in rows.items():
    print(f"{name:46s}", "  ".join(f"{term}: " + " ".join(f"{abs(losses[k][term] - base[k][term]) / (abs(base[k][term]) + 1e-12):.1e}"
                                                       for k in (3, 6, 9)) for term in terms if term in base[0]))
