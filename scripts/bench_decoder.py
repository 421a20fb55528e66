"""The RGB CNN decoder standalone: forward + backward of decode_rgb on the c3 batch's 40 patches of
32 x 32 features under the settings MIOpen offers (autocast dtype, channels_last weights, find mode) + the kernels behind
the best one.   python scripts/bench_decoder.py"""
import itertools
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from neurad_studio_amd.model_components.cnns import decode_rgb, make_rgb_decoder

dev = torch.device("cuda")
n_cam = 40960
image = torch.rand((n_cam // 1024, 96, 96, 3), device=dev)


def run(dtype, cl, find, n=12):
    torch.backends.cudnn.benchmark = find
    torch.manual_seed(0)
    dec = make_rgb_decoder(48, 32, 3).to(dev).train()
    if cl:
        dec = dec.to(memory_format=torch.channels_last)
    f48 = torch.randn((n_cam, 48), device=dev, requires_grad=True)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for k in range(n + 6):
        if k >= 6:
            ev[k - 6][0].record()
        with torch.autocast("cuda", dtype=dtype, enabled=dtype is not None):
            rgb = decode_rgb(dec, f48, (32, 32))
        torch.nn.functional.mse_loss(rgb.float(), image).backward()
        if k >= 6:
            ev[k - 6][1].record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)
    return ms[len(ms) // 2], dec, f48


res = {}
for dtype, cl, find in itertools.product((torch.float16, torch.bfloat16, None), (False, True), (False, True)):
    try:
        ms, _, _ = run(dtype, cl, find)
    except Exception as e:  # noqa: BLE001
        ms = f"{type(e).__name__}: {e}"[:120]
    res[f"{str(dtype).replace('torch.', '')}|channels_last={cl}|find={find}"] = ms
    print(list(res.items())[-1], flush=True)
best = min((v, k) for k, v in res.items() if isinstance(v, float))
print(json.dumps({"fwd_bwd_ms": res, "best": best}))
dt, cl, find = best[1].split("|")
dtype = {"float16": torch.float16, "bfloat16": torch.bfloat16, "None": None}[dt]
cl, find = cl.endswith("True"), find.endswith("True")
ms, dec, f48 = run(dtype, cl, find, n=3)
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) as prof:
    for _ in range(3):
        with torch.autocast("cuda", dtype=dtype, enabled=dtype is not None):
            rgb = decode_rgb(dec, f48, (32, 32))
        torch.nn.functional.mse_loss(rgb.float(), image).backward()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=90))
