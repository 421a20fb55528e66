"""RGB decoder forward + backward at the c3 step's shape (40 patches of 32x32 -> 96x96): HIP kernels vs the torch modules
under fp16 autocast (MIOpen).  Prints one JSON object."""
import copy
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurad_studio_amd.model_components.cnns import decode_rgb, make_rgb_decoder  # noqa: E402


def main():
    torch.manual_seed(0)
    dec = make_rgb_decoder(48, 32, 3).cuda().train()
    B = 40
    f = torch.randn((B * 1024, 48), device="cuda", requires_grad=True)
    image = torch.rand((B, 96, 96, 3), device="cuda")
    res = {}

    def step(mode, d):
        for p in d.parameters():  # as optimizer.zero_grad(set_to_none=True) does: without it every parameter gradient costs an
            p.grad = None         # accumulation launch (36 torch adds per pass in round 3's trace: this script's artefact)
        f.grad = None
        if mode == "hip":
            rgb = decode_rgb(d, f, (32, 32))
        else:
            with torch.autocast("cuda", dtype=torch.float16):
                rgb = decode_rgb(d, f, (32, 32), fused=False)
        torch.nn.functional.mse_loss(rgb.float(), image).backward()

    for mode in os.environ.get("NRHIP_BENCH_DECODER_MODES", "hip,miopen_autocast_fp16").split(","):
        d = copy.deepcopy(dec)
        for _ in range(3):
            step(mode, d)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
        for a, b in ev:
            a.record()
            step(mode, d)
            b.record()
        torch.cuda.synchronize()
        t = sorted(a.elapsed_time(b) for a, b in ev)
        res[mode + "_fwd_bwd_ms"] = round(t[len(t) // 2], 3)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
