"""A/B of the fused kernels' variants on the GPU (HIP-event timing, no profiler):

  composited render (nrhip_render_fwd_ex, variant 1/2/3, + early ray termination) on BASELINE config[1] and on
  NeuRAD's default grid, and -- in one subprocess per NRHIP_RENDER_VARIANT value, because the per-sample entry points
  pick their kernel from the environment -- field_fwd (eval) and field_fwd_train (training forward).

    python scripts/bench_render_variants.py            # everything
    python scripts/bench_render_variants.py --child    # (internal) per-sample paths under the current environment
"""
import json
import os
import subprocess
import sys

sys.path.insert(0, os.getcwd())
sys.path.insert(0, "tests")
import numpy as np
import torch

import synth
from neurad_studio_amd import ops

dev = lambda a, dt=torch.float32: torch.from_numpy(np.ascontiguousarray(a)).to("cuda", dt)  # noqa: E731


def lin(o, i, s):
    w, b = synth.linear(o, i, s)
    return dev(w), dev(b)


def mk(L, F, lg, H, mn, mx):
    spec = ops.GridSpec(L, F, lg, mn, mx)
    table = (torch.rand(L * 2**lg, F, device="cuda") * 2 - 1) * 1e-3
    gw0, gb0 = lin(H, 32, 1)
    gw1, gb1 = lin(33, H, 2)
    fw0, fb0 = lin(H, 48, 3)
    fw1, fb1 = lin(H, H, 4)
    fw2, fb2 = lin(32, H, 5)
    return ops.FieldSpec(spec, table, 100.0, [gw0, gw1], [gb0, gb1], [fw0, fw1, fw2], [fb0, fb1, fb2], True, 20.0001)


def timeit(fn, iters=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us


CFGS = {"config1_16x2_T19_H64_4096x128": (16, 2, 19, 64, 16, 1024, 4096, 128),
        "neurad_8x4_T22_H32_16384x32": (8, 4, 22, 32, 32, 8192, 16384, 32),
        "neurad_8x4_T22_H32_4096x128": (8, 4, 22, 32, 32, 8192, 4096, 128)}


def workload(cfg):
    L, F, lg, H, mn, mx, R, S = cfg
    fs = mk(L, F, lg, H, mn, mx)
    g = torch.Generator(device="cuda")
    g.manual_seed(1234)
    o = torch.randn((R, 3), device="cuda", generator=g) * 5.0
    d = torch.randn((R, 3), device="cuda", generator=g)
    d = d / d.norm(dim=-1, keepdim=True)
    area = torch.full((R,), 2.43e-6, device="cuda")
    eu = ops.power_sampler(None, torch.full((R,), 20000.0, device="cuda"), S, last_edge=20000.0)[1]
    return fs, o.contiguous(), d.contiguous(), area, eu


def child():
    out = {}
    for name, cfg in CFGS.items():
        fs, o, d, area, eu = workload(cfg)
        out[name] = {
            "field_fwd_us": timeit(lambda: ops.field_fwd(fs, o, d, area, eu[:, :-1], eu[:, 1:])),
            "field_fwd_train_us": timeit(lambda: ops.field_fwd_train(fs, o, d, area, eu[:, :-1], eu[:, 1:]), iters=15),
        }
    print("CHILD " + json.dumps(out))


def main():
    res = {}
    for name, cfg in CFGS.items():
        fs, o, d, area, eu = workload(cfg)
        R, S = cfg[6], cfg[7]
        outs = {}
        row = {}
        for v in (1, 2, 3):
            row[f"render_v{v}_us"] = timeit(lambda: ops.render_fwd(fs, o, d, area, eu[:, :-1], eu[:, 1:], variant=v))
            outs[v] = ops.render_fwd(fs, o, d, area, eu[:, :-1], eu[:, 1:], variant=v)
        for v in (2, 3):
            row[f"rel_l2_v{v}_vs_v1"] = float((outs[v][0] - outs[1][0]).norm() / outs[1][0].norm())
        for eps in (1e-4, 1e-2):
            row[f"render_v3_stop{eps:g}_us"] = timeit(
                lambda: ops.render_fwd(fs, o, d, area, eu[:, :-1], eu[:, 1:], variant=3, early_stop_eps=eps))
            fe = ops.render_fwd(fs, o, d, area, eu[:, :-1], eu[:, 1:], variant=3, early_stop_eps=eps)[0]
            row[f"max_abs_err_stop{eps:g}"] = float((fe - outs[3][0]).abs().max())
        row["samples"] = R * S
        res[name] = row
        print(name, json.dumps(row), flush=True)
        del fs
        torch.cuda.empty_cache()
    for v in ("1", "2"):
        env = dict(os.environ, NRHIP_RENDER_VARIANT=v)
        p = subprocess.run([sys.executable, __file__, "--child"], env=env, capture_output=True, text=True)
        line = [l for l in p.stdout.splitlines() if l.startswith("CHILD ")]
        if not line:
            print("child failed:", p.stderr[-2000:])
            continue
        for name, r in json.loads(line[0][6:]).items():
            for k, val in r.items():
                res[name][f"{k[:-3]}_v{v}_us"] = val
    print("RESULT " + json.dumps(res))


if __name__ == "__main__":
    child() if "--child" in sys.argv else main()
