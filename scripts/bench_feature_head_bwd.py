"""Feature-head backward: the one-pass kernel (nrhip_field_feature_bwd) against the composition it replaces
(nrhip_mlp_bwd + column copy + residual add), then the geometry MLP backward that consumes the result, at the c1 and c3
sizes.   python scripts/bench_feature_head_bwd.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurad_studio_amd import ops  # noqa: E402


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    res = {"direct_stores": os.environ.get("NRHIP_EXP_RES_DIRECT") is not None}
    for name, N, H in (("c1", 4096 * 128, 64), ("c3", 57344 * 32, 32)):
        g = torch.Generator(device="cuda").manual_seed(0)
        mk = lambda *s: torch.randn(s, device="cuda", generator=g)  # noqa: E731
        fw = [mk(H, 48) * 0.2, mk(H, H) * 0.2, mk(32, H) * 0.2]
        fb = [mk(H) * 0.1, mk(H) * 0.1, mk(32) * 0.1]
        gw = [mk(H, 32) * 0.2, mk(33, H) * 0.2]
        gb = [mk(H) * 0.1, mk(33) * 0.1]
        enc, xf = mk(N, 32), mk(N, 48)
        _, hf = ops.mlp_fwd(xf, fw, fb, save_hidden=True)
        _, hg = ops.mlp_fwd(enc, gw, gb, save_hidden=True)
        gfeat, g0 = mk(N, 32), mk(N)

        def composed():
            gxf, a, b = ops.mlp_bwd(xf, hf, gfeat, fw, fb)
            g_geo = torch.empty((N, 33), device="cuda")
            g_geo[:, 0] = g0
            torch.add(gfeat, gxf[:, :32], out=g_geo[:, 1:])
            return g_geo

        def one_pass():
            return ops.field_feature_bwd(xf, hf, gfeat, g0, fw, fb)[0]

        assert (composed() - one_pass()).abs().max() < 1e-4
        res[name] = {"composed_us": timeit(composed), "one_pass_us": timeit(one_pass),
                     "composed_then_geo_us": timeit(lambda: ops.mlp_bwd(enc, hg, composed(), gw, gb)),
                     "one_pass_then_geo_us": timeit(lambda: ops.mlp_bwd(enc, hg, one_pass(), gw, gb))}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
