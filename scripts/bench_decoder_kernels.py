"""Microbenchmark of the decoder kernels at the c3 step's shapes (40 patches: 32x32 and 96x96, 32 channels)."""
import json
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurad_studio_amd import ops_decoder as D  # noqa: E402


def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in ev)
    return t[len(t) // 2] * 1e3  # us (median)


def main():
    res = {}
    if "--profile" in sys.argv:  # few launches of the two big kernels, for counter passes
        x = torch.randn((40, 96, 96, 32), device="cuda").half()
        wf = D.conv7x7_pack(torch.randn((32, 32, 7, 7), device="cuda") * 0.05, 0)
        gw, gb = torch.zeros((32, 32, 7, 7), device="cuda"), torch.zeros((32,), device="cuda")
        for _ in range(5):
            D.conv7x7(x, wf, gb, stats=True, rows_per_wave=4)
            D.conv7x7_wgrad(x, x, gw, gb)
        torch.cuda.synchronize()
        return
    w = torch.randn((32, 32, 7, 7), device="cuda") * 0.05
    wf = D.conv7x7_pack(w, 0)
    bias = torch.zeros(32, device="cuda")
    for (B, H, W) in ((40, 32, 32), (40, 96, 96)):
        x = torch.randn((B, H, W, 32), device="cuda").half()
        flops = 2.0 * B * H * W * 49 * 32 * 32
        for r in (1, 2, 4):
            for st in (False, True):
                us = timeit(lambda: D.conv7x7(x, wf, bias, stats=st, rows_per_wave=r))
                res[f"conv7x7 {B}x{H}x{W} R={r} stats={int(st)}"] = {"us": round(us, 1), "TFLOPs": round(flops / us / 1e6, 1)}
        gw = torch.zeros((32, 32, 7, 7), device="cuda")
        gb = torch.zeros((32,), device="cuda")
        us = timeit(lambda: D.conv7x7_wgrad(x, x, gw, gb))
        res[f"conv7x7_wgrad {B}x{H}x{W}"] = {"us": round(us, 1), "TFLOPs": round(flops / us / 1e6, 1)}
        xc = x.permute(0, 3, 1, 2).contiguous(memory_format=torch.channels_last)
        wh = w.half().contiguous(memory_format=torch.channels_last)
        us = timeit(lambda: torch.nn.functional.conv2d(xc, wh, None, padding=3))
        res[f"MIOpen conv2d fp16 channels_last {B}x{H}x{W}"] = {"us": round(us, 1), "TFLOPs": round(flops / us / 1e6, 1)}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
