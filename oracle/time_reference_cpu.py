"""Time the REFERENCE's own torch field-eval path on this machine's host cores (build container only; the GPU box has no
reference tree) -> profiles/reference_torch_cpu.json, which bench.py reports as `reference_torch_cpu` next to the C port.

Workload = BASELINE config[1]: NeuRADField(implementation="torch") with HashEncoding(16 levels, T=2^19, F=2) and 64-wide
MLPs on 4096 rays x 128 PowerSampler samples, forward under no_grad, median of 5 after 2 warm-ups;
then forward + backward of the same batch.   python oracle/time_reference_cpu.py"""
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
import torch

import ref_import

ref_import.install()
from nerfstudio.cameras.rays import RayBundle  # noqa: E402
from nerfstudio.field_components.field_heads import FieldHeadNames  # noqa: E402
from nerfstudio.field_components.neurad_encoding import NeuRADHashEncodingConfig, StaticSettings  # noqa: E402
from nerfstudio.fields.neurad_field import NeuRADField, NeuRADFieldConfig  # noqa: E402
from nerfstudio.model_components.dynamic_actors import DynamicActors, DynamicActorsConfig  # noqa: E402
from nerfstudio.model_components.ray_samplers import PowerSampler  # noqa: E402

torch.set_num_threads(os.cpu_count())
torch.manual_seed(0)
R, S = 4096, 128
grid = NeuRADHashEncodingConfig(static=StaticSettings(hashgrid_dim=2, num_levels=16, base_res=16, max_res=1024,
                                                      log2_hashmap_size=19))
cfg = NeuRADFieldConfig(grid=grid, geo_hidden_dim=64, nff_hidden_dim=64)
fld = NeuRADField(cfg, actors=DynamicActors(DynamicActorsConfig(), trajectories=[]), static_scale=100.0,
                  implementation="torch").eval()
o = torch.randn(R, 3) * 5
d = torch.nn.functional.normalize(torch.randn(R, 3), dim=-1)
rb = RayBundle(origins=o, directions=d, pixel_area=torch.full((R, 1), 2.43e-6), nears=torch.zeros(R, 1),
               fars=torch.full((R, 1), 20000.0), times=torch.zeros(R, 1))
rs = PowerSampler(num_samples=S, lambda_=-1.0, scaling=0.1).eval()(rb)


def fwd():
    with torch.no_grad():
        return fld(rs)


def fwd_bwd():
    out = fld(rs)
    (out[FieldHeadNames.FEATURE].square().mean() + out[FieldHeadNames.ALPHA].mean()).backward()
    fld.zero_grad(set_to_none=True)


def med(fn, n=5, warm=2):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2]


tf, tb = med(fwd), med(fwd_bwd, n=3, warm=1)
out = {"what": "reference NeuRADField(implementation='torch'), BASELINE config[1] grid (16 levels, T=2^19, F=2, 64-wide), "
               "4096 rays x 128 samples, fp32, torch CPU ops", "where": "build container", "cores": os.cpu_count(),
       "torch_threads": torch.get_num_threads(), "forward_s": tf, "forward_ray_samples_per_s": R * S / tf,
       "forward_backward_s": tb, "forward_backward_ray_samples_per_s": R * S / tb}
if "--no-write" in sys.argv:  # bench.py's live timing on a host that has the reference tree: stdout only
    out["where"] = "this host (live)"
else:
    json.dump(out, open(os.path.join(ROOT, "profiles", "reference_torch_cpu.json"), "w"), indent=1)
print(json.dumps(out))
