"""Dynamic-actor scene (H5) at a realistic size: NeuRAD default grids (static L=8 F=4 T=2^22, actor grids L=4 F=4
T=2^17), N_ACTORS moving boxes with 50-pose trajectories, 16384 rays x 32 samples with per-ray times.
Prints forward (eval) and forward+backward (train) times of NeuRADField on the operator-level actor path.

  python scripts/bench_actors.py [n_actors] [n_rays]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from neurad_studio_amd.cameras.rays import RayBundle
from neurad_studio_amd.field_components.field_heads import FieldHeadNames
from neurad_studio_amd.fields.neurad_field import NeuRADField, NeuRADFieldConfig
from neurad_studio_amd.model_components.dynamic_actors import DynamicActors, DynamicActorsConfig
from neurad_studio_amd.model_components.ray_samplers import PowerSampler

NA = int(sys.argv[1]) if len(sys.argv) > 1 else 20
R = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
S = 32
dev = torch.device("cuda")
rng = np.random.default_rng(0)
trajs = []
ts = torch.linspace(0.0, 5.0, 50)
for a in range(NA):
    x0, y0, yaw, v = rng.uniform(-40, 40), rng.uniform(-40, 40), rng.uniform(-3, 3), rng.uniform(0, 8)
    poses = []
    for t in ts:
        c, s = np.cos(yaw), np.sin(yaw)
        p = torch.eye(4)
        p[:3, :3] = torch.tensor([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]], dtype=torch.float32)
        p[:3, 3] = torch.tensor([x0 + v * float(t) * c, y0 + v * float(t) * s, 0.8], dtype=torch.float32)
        poses.append(p)
    trajs.append({"timestamps": ts.clone(), "poses": torch.stack(poses), "dims": torch.tensor([2.0, 4.6, 1.6]),
                  "symmetric": torch.tensor(True), "deformable": torch.tensor(False)})
actors = DynamicActors(DynamicActorsConfig(), trajectories=trajs)
fld = NeuRADField(NeuRADFieldConfig(), actors=actors, static_scale=100.0).to(dev)
with torch.no_grad():
    fld.hashgrid.static_grid.hash_table.mul_(1000.0)
g = torch.Generator(device=dev)
g.manual_seed(1)
o = torch.randn((R, 3), device=dev, generator=g) * 20.0
o[:, 2] = 1.5
d = torch.randn((R, 3), device=dev, generator=g)
d[:, 2] *= 0.1
d = d / d.norm(dim=-1, keepdim=True)
times = torch.rand((R, 1), device=dev, generator=g) * 5.0
rb = RayBundle(origins=o, directions=d, pixel_area=torch.full((R, 1), 2.43e-6, device=dev),
               nears=torch.zeros((R, 1), device=dev), fars=torch.full((R, 1), 150.0, device=dev), times=times)
sampler = PowerSampler(num_samples=S, lambda_=-1.0, scaling=0.1).to(dev).eval()


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def fwd():
    with torch.no_grad():
        return fld.eval()(sampler(rb))


def train():
    out = fld.train()(sampler(rb))
    loss = out[FieldHeadNames.FEATURE].square().mean() + out[FieldHeadNames.ALPHA].mean()
    for p in fld.parameters():
        p.grad = None
    loss.backward()


def train_fused():  # the fused training node: field + head + compositing, actor rows as overrides (nrhip_field_fwd_train_ovr)
    eu = torch.cat([sampler(rb).frustums.starts[..., 0], rb.fars], -1)
    feats, depth, acc, w = fld.train().render_train(o, d, rb.pixel_area, eu, times=times)
    loss = feats.square().mean() + acc.mean() + 1e-3 * depth.mean()
    for p in fld.parameters():
        p.grad = None
    loss.backward()


if len(sys.argv) > 3 and sys.argv[3] in ("train_op", "train_fused"):  # profile targets: one path alone
    fn = train if sys.argv[3] == "train_op" else train_fused
    print(f"{NA} actors, {R} rays x {S} samples: {sys.argv[3]} forward+backward {timeit(fn, 20):.3f} ms")
    sys.exit(0)
if len(sys.argv) > 3 and sys.argv[3] == "train":  # profile target: the training step alone
    print(f"{NA} actors, {R} rays x {S} samples: train forward+backward {timeit(train, 20):.3f} ms (operator level), "
          f"{timeit(train_fused, 20):.3f} ms (fused node with row overrides)")
    sys.exit(0)
rs = sampler(rb)
fr = rs.frustums
starts, ends = fr.starts[..., 0].contiguous(), fr.ends[..., 0].contiguous()
pa, tm = rb.pixel_area[:, 0].contiguous(), times[:, 0].contiguous()


def fused():  # nrhip_actor_prepare + device-side ray split + static kernel + actor kernel
    return fld.eval().render(o, d, pa, starts, ends, times=tm)


def fused_static_only():  # the same rays through the static kernel alone (no actors): the floor
    from neurad_studio_amd import ops
    return ops.render_fwd(fld.field_spec(), o, d, pa, starts, ends)


with torch.no_grad():
    out = fld.eval()(rs)
    cnt = fld.hashgrid.prepare_actors(o, d, pa, starts, ends, tm)[1][0]
    from neurad_studio_amd import ops
    from neurad_studio_amd.shims import nerfacc
    w = nerfacc.render_weight_from_alpha(out[FieldHeadNames.ALPHA][..., 0])[0]
    acc_ref = w.sum(-1, keepdim=True)
    f, dep, acc = fused()
    hit = ops.actor_hits(*fld.hashgrid.prepare_actors(o, d, pa, starts, ends, tm), o, d, pa, starts, ends)
print(f"{NA} actors, {R} rays x {S} samples: {float((cnt > 0).float().mean()) * 100:.1f}% of the rays have candidate actors "
      f"(max {int(cnt.max())}), {float((hit[:, 0] >= 0).float().mean()) * 100:.2f}% of the samples lie in a box")
print(f"  fused vs operator path: accumulation max abs diff {float((acc - acc_ref).abs().max()):.2e}")
print(f"  eval, operator-level path (per-sample outputs, no compositing) {timeit(fwd):.3f} ms")
print(f"  eval, fused render with actors (prepare + split + 2 kernels)    {timeit(fused):.3f} ms")
print(f"  eval, fused render of the static scene alone (floor)            {timeit(fused_static_only):.3f} ms")
print(f"  train forward+backward (operator-level actor path)             {timeit(train):.3f} ms")
print(f"  train forward+backward (fused node, actor rows as overrides)    {timeit(train_fused):.3f} ms")
