"""tests/golden/losses.npz: the reference's own zipnerf_interlevel_loss and distortion_loss (model_components/losses.py),
forward values and autograd gradients, on the sampler chain NeuRADModel drives (same inputs as make_golden.py's
golden_sampler, 20 rays, 128 -> 64 -> 32 samples).  Build container only:  python oracle/make_golden_losses.py"""
import os
import sys
from types import SimpleNamespace

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np
import torch

import ref_import

ref_import.install()
import synth  # noqa: E402
from nerfstudio.model_components.losses import distortion_loss, zipnerf_interlevel_loss  # noqa: E402

T = torch.from_numpy
g = np.load(os.path.join(ROOT, "tests", "golden", "sampler_chain.npz"))
R = g["w0"].shape[0]


def fake_samples(starts, ends):
    """ray_samples_to_sdist (losses.py:107-112) reads only spacing_starts / spacing_ends"""
    return SimpleNamespace(spacing_starts=T(np.ascontiguousarray(starts))[..., None],
                           spacing_ends=T(np.ascontiguousarray(ends))[..., None])


def sdist_of(euclid_starts, euclid_ends, fars):
    """the chain golden stores euclidean bins for the proposal rounds; spacing bins through PowerSampler's own map"""
    from nerfstudio.model_components.ray_samplers import PowerSampler

    ps = PowerSampler(lambda_=-1.0, scaling=0.1)
    near, far = torch.zeros(R, 1), T(np.minimum(fars, 20000.0).astype(np.float32))[:, None]
    s_near, s_far = ps.spacing_fn(near), ps.spacing_fn(far)
    edges = torch.cat([T(euclid_starts), T(euclid_ends)[:, -1:]], -1)
    return ((ps.spacing_fn(edges) - s_near) / (s_far - s_near)).clamp(0, 1)


sd0, sd1 = sdist_of(g["s0"], g["e0"], g["fars"]), sdist_of(g["s1"], g["e1"], g["fars"])
sdf = torch.cat([T(g["sps"]), T(g["spe"])[:, -1:]], -1)
rsl = [fake_samples(sd0[:, :-1].numpy(), sd0[:, 1:].numpy()), fake_samples(sd1[:, :-1].numpy(), sd1[:, 1:].numpy()),
       fake_samples(g["sps"], g["spe"])]
# final (field) weights: the final bins are PDF samples of round 1, i.e. equal-mass bins of its histogram -> field
# weights of a model whose proposals are roughly right: that mass, perturbed per sample and scaled per ray
Sf = g["sps"].shape[1]
wf = g["w1"].sum(-1, keepdims=True) / Sf * synth.uniform((R, Sf), 0.3, 1.9, seed=501)
wf = wf * synth.uniform((R, 1), 0.6, 1.0, seed=502)
w0 = T(g["w0"].copy()).requires_grad_(True)
w1 = T(g["w1"].copy()).requires_grad_(True)
wl = [w0[..., None], w1[..., None], T(wf.astype(np.float32)).requires_grad_(True)[..., None]]
il = zipnerf_interlevel_loss(wl, rsl)
il.backward()
wfin = T(wf.astype(np.float32)).requires_grad_(True)
dl = distortion_loss([wfin[..., None]], [rsl[-1]])
dl.backward()
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "losses.npz"), sd0=sd0.numpy(), sd1=sd1.numpy(), sdf=sdf.numpy(),
                    w0=g["w0"], w1=g["w1"], wf=wf.astype(np.float32), interlevel=il.detach().numpy(),
                    g_w0=w0.grad.numpy(), g_w1=w1.grad.numpy(), distortion=dl.detach().numpy(), g_wf=wfin.grad.numpy())
print("interlevel", float(il), "distortion", float(dl), "|g_w0|", float(w0.grad.abs().sum()), "|g_w1|", float(w1.grad.abs().sum()))
