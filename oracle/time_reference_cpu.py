"""Time the REFERENCE's own torch field-eval path on this machine's host cores (the tree ref_import finds: /root/reference
in the build container, the byte-compiled oracle/_ref on the GPU box).  bench.py runs it live (--no-write) as its
`cpu_baseline`; without the flag it records profiles/reference_torch_cpu.json.

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

def _arg(flag, default):
    return type(default)(sys.argv[sys.argv.index(flag) + 1]) if flag in sys.argv else default


LIVE = "--no-write" in sys.argv  # bench.py's bounded live leg: a slice of the batch, a few thread counts, ~30 s in all
torch.manual_seed(0)
R, S = _arg("--rays", 512 if LIVE else 4096), 128
DEFAULT_THREADS = torch.get_num_threads()
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


def med(fn, n=5, warm=2, budget_s=1e9):
    for _ in range(warm):
        fn()
    ts, t_all = [], time.perf_counter()
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
        if time.perf_counter() - t_all > budget_s:
            break
    return sorted(ts)[len(ts) // 2]


# torch's CPU gather path does not scale with threads (256 threads on a 128-core host ran 14x SLOWER than 8 in the build
# container): the reference gets the best of a few thread counts, the count is reported
tried = {}
cands = [DEFAULT_THREADS] + [t for t in (64, 32, 16, 8) if t < DEFAULT_THREADS] if LIVE else [os.cpu_count()]
for nt in cands:
    torch.set_num_threads(nt)
    tried[nt] = med(fwd, n=3 if LIVE else 5, warm=1 if LIVE else 2, budget_s=5.0 if LIVE else 1e9)
best = min(tried, key=tried.get)
torch.set_num_threads(best)
tf = tried[best]
tb = med(fwd_bwd, n=2 if LIVE else 3, warm=1, budget_s=6.0 if LIVE else 1e9)
out = {"what": "reference NeuRADField(implementation='torch'), BASELINE config[1] grid (16 levels, T=2^19, F=2, 64-wide), "
               f"{R} rays x {S} samples per pass, fp32, torch CPU ops", "where": "build container", "cores": os.cpu_count(),
       "rays": R, "samples": S, "torch_threads": best, "torch_threads_default": DEFAULT_THREADS,
       "forward_s_by_threads": {str(k): v for k, v in tried.items()}, "forward_s": tf,
       "forward_ray_samples_per_s": R * S / tf, "forward_backward_s": tb, "forward_backward_ray_samples_per_s": R * S / tb}
if LIVE:  # bench.py's live timing: stdout only
    out["where"] = "this host (live)"
else:
    json.dump(out, open(os.path.join(ROOT, "profiles", "reference_torch_cpu.json"), "w"), indent=1)
print(json.dumps(out))
