"""Which torch ops make up the library-kernel share of the c3 training step?  torch.profiler over 3 steps, aten ops by
device time with their input shapes.   python scripts/c3_glue_profile.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile

import bench

dev = torch.device("cuda:0")
steps = {}
orig_timed = bench.timed


def timed(step, n, warmup, world, device):
    for _ in range(4):
        step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        for _ in range(3):
            step()
        torch.cuda.synchronize()
    steps["prof"] = prof
    return orig_timed(step, n, warmup, world, device)


bench.timed = timed
bench.train_full_section(dev, 0, 1, 3, 1)
ka = steps["prof"].key_averages(group_by_input_shape=True)
rows = sorted(ka, key=lambda e: -getattr(e, "self_device_time_total", getattr(e, "self_cuda_time_total", 0)))
print("op | calls/step | device us/step | shapes")
tot = 0.0
only = sys.argv[1] if len(sys.argv) > 1 else ""
rows = [e for e in rows if e.key.startswith(only)] if only else rows
for e in rows[:int(os.environ.get("GLUE_ROWS", "45"))]:
    t = getattr(e, "self_device_time_total", getattr(e, "self_cuda_time_total", 0)) / 3
    tot += t
    print(f"{e.key[:38]:38s} {e.count / 3:6.1f} {t:9.1f}  {str(e.input_shapes)[:110]}")
