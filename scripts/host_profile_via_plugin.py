"""Where the HOST time of the `--via-plugin` c3 step goes (it is host-bound once the trainer's get_scale() reads are gone):
cProfile over the timed steps of bench.train_via_plugin_section, cumulative and own time per function.
usage (GPU box): python scripts/host_profile_via_plugin.py [steps] > gpurun_out/host_profile.txt"""
import cProfile
import io
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
real_timed = bench.timed
prof = cProfile.Profile()
calls = []


def timed(step, n, warmup, world, device):
    if calls:  # (the second timed() of the section is the reference trainer's leg: not profiled)
        return real_timed(step, n, warmup, world, device)
    calls.append(1)
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    prof.enable()
    for i in range(n):
        step(i)
    prof.disable()
    torch.cuda.synchronize()
    return real_timed(step, n, 0, world, device)


bench.timed = timed
out = bench.train_via_plugin_section(dev, 0, 1, steps, 5)
print("ms_per_iter", out["ms_per_iter"], "host_enqueue_ms_per_step", out["host_enqueue_ms_per_step"])
for key in ("cumulative", "tottime"):
    s = io.StringIO()
    pstats.Stats(prof, stream=s).strip_dirs().sort_stats(key).print_stats(45)
    print(f"==== by {key} (totals over {steps} steps) ====")
    print(s.getvalue())
