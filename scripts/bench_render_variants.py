"""Fused field/render kernels on the GPU (HIP-event timing, no profiler): batch order as given vs the processing order
from nrhip_ray_order, on random rays and on 32x32 camera patches; early ray termination; the ordering pass itself.

    python scripts/bench_render_variants.py
"""
import json
import os
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
    return round(e0.elapsed_time(e1) / iters * 1e3, 1)  # us


CFGS = {"config1_16x2_T19_H64_4096x128": (16, 2, 19, 64, 16, 1024, 4096, 128),
        "neurad_8x4_T22_H32_16384x32": (8, 4, 22, 32, 32, 8192, 16384, 32),
        "neurad_8x4_T22_H32_4096x128": (8, 4, 22, 32, 32, 8192, 4096, 128)}


def rays(R, mode, g):
    if mode == "patches":  # R/1024 cameras, one 32x32 pixel patch each (pinhole f=1900 px, 1920x1280)
        n = R // 1024
        o = (torch.randn((n, 1, 3), device="cuda", generator=g) * 5.0).expand(n, 1024, 3).reshape(R, 3)
        fwd = torch.randn((n, 3), device="cuda", generator=g)
        fwd = fwd / fwd.norm(dim=-1, keepdim=True)
        up = torch.tensor([0.0, 0.0, 1.0], device="cuda").expand(n, 3)
        right = torch.cross(fwd, up, dim=-1)
        right = right / right.norm(dim=-1, keepdim=True)
        up2 = torch.cross(right, fwd, dim=-1)
        u0 = torch.randint(0, 1920 - 32, (n,), device="cuda", generator=g)
        v0 = torch.randint(0, 1280 - 32, (n,), device="cuda", generator=g)
        vv, uu = torch.meshgrid(torch.arange(32, device="cuda"), torch.arange(32, device="cuda"), indexing="ij")
        x = ((u0[:, None, None] + uu[None]) - 960.0) / 1900.0
        y = ((v0[:, None, None] + vv[None]) - 640.0) / 1900.0
        d = fwd[:, None, None, :] + x[..., None] * right[:, None, None, :] - y[..., None] * up2[:, None, None, :]
        d = d.reshape(R, 3)
    else:
        o = torch.randn((R, 3), device="cuda", generator=g) * 5.0
        d = torch.randn((R, 3), device="cuda", generator=g)
    d = d / d.norm(dim=-1, keepdim=True)
    return o.contiguous(), d.contiguous()


def main():
    res = {}
    for name, cfg in CFGS.items():
        L, F, lg, H, mn, mx, R, S = cfg
        fs = mk(L, F, lg, H, mn, mx)
        area = torch.full((R,), 2.43e-6, device="cuda")
        eu = ops.power_sampler(None, torch.full((R,), 20000.0, device="cuda"), S, last_edge=20000.0)[1]
        st, en = eu[:, :-1], eu[:, 1:]
        for mode in ("random", "patches"):
            g = torch.Generator(device="cuda")
            g.manual_seed(1234)
            o, d = rays(R, mode, g)
            order = ops.ray_order(o, d, 100.0)
            row = {"ray_order_us": timeit(lambda: ops.ray_order(o, d, 100.0)),
                   "render_us": timeit(lambda: ops.render_fwd(fs, o, d, area, st, en)),
                   "render_ordered_us": timeit(lambda: ops.render_fwd(fs, o, d, area, st, en, order=order)),
                   "order+render_us": timeit(lambda: ops.render_fwd(fs, o, d, area, st, en,
                                                                    order=ops.ray_order(o, d, 100.0))),
                   "field_fwd_us": timeit(lambda: ops.field_fwd(fs, o, d, area, st, en)),
                   "field_fwd_ordered_us": timeit(lambda: ops.field_fwd(fs, o, d, area, st, en, order=order)),
                   "field_fwd_train_us": timeit(lambda: ops.field_fwd_train(fs, o, d, area, st, en), iters=15),
                   "field_fwd_train_ordered_us": timeit(lambda: ops.field_fwd_train(fs, o, d, area, st, en, order=order),
                                                        iters=15)}
            for tr in (20.0, 300.0):
                od = ops.ray_order(o, d, 100.0, t_ref=tr)
                row[f"render_ordered_tref{tr:g}_us"] = timeit(lambda: ops.render_fwd(fs, o, d, area, st, en, order=od))
            for kb in (4, 5):
                od = ops.ray_order(o, d, 100.0, key_bits=kb)
                row[f"ray_order_bits{kb}_us"] = timeit(lambda: ops.ray_order(o, d, 100.0, key_bits=kb))
                row[f"render_ordered_bits{kb}_us"] = timeit(lambda: ops.render_fwd(fs, o, d, area, st, en, order=od))
            exact = ops.render_fwd(fs, o, d, area, st, en)[0]
            for eps in (1e-4, 1e-2):
                row[f"render_stop{eps:g}_us"] = timeit(lambda: ops.render_fwd(fs, o, d, area, st, en, early_stop_eps=eps))
                fe = ops.render_fwd(fs, o, d, area, st, en, early_stop_eps=eps)[0]
                row[f"max_abs_err_stop{eps:g}"] = float((fe - exact).abs().max())
            res[f"{name}/{mode}"] = row
            print(f"{name}/{mode}", json.dumps(row), flush=True)
        del fs
        torch.cuda.empty_cache()
    print("RESULT " + json.dumps(res))


if __name__ == "__main__":
    main()
