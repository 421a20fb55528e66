"""queue fill of the binned table-gradient path on the bench workload (records per slice, per level)"""
import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
import torch
from neurad_studio_amd import ops
from neurad_studio_amd._lib import call
R, S = 4096, 128
for (L, F, lg, mn, mx) in [(16, 2, 19, 16, 1024), (8, 4, 22, 32, 8192)]:
    spec = ops.GridSpec(L, F, lg, mn, mx)
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    o = torch.randn((R, 3), device="cuda", generator=g) * 5
    d = torch.randn((R, 3), device="cuda", generator=g); d = d / d.norm(dim=-1, keepdim=True)
    area = torch.full((R,), 2.43e-6, device="cuda")
    sp, eu = ops.power_sampler(None, torch.full((R,), 20000.0, device="cuda"), S)
    go = torch.randn((R * S, L * F), device="cuda", generator=g)
    r, keep = ops._c_rays(o, d, area, eu[:, :-1], eu[:, 1:])
    gt = torch.zeros((spec.table_rows, F), device="cuda")
    cg = spec.c_grid(gt)
    need = C.c_int64(0)
    call("nrhip_encode_bwd_binned_workspace", C.byref(cg), R * S, C.byref(need))
    ws = torch.empty((need.value,), device="cuda", dtype=torch.uint8)
    call("nrhip_encode_bwd_binned", C.byref(cg), 100.0, C.byref(r), ops._ptr(go), ops._ptr(gt), ops._ptr(ws), need.value, ops._stream())
    torch.cuda.synchronize()
    log2ts = min(14 - (F.bit_length() - 1), lg)
    while (L << (lg - log2ts)) < 512 and log2ts > 9:
        log2ts -= 1
    nb = 1 << (lg - log2ts)
    cnt = ws[: L * nb * 4].view(torch.int32).view(L, nb).double()
    cap = 2 * ((R * S * 8 + nb - 1) // nb) + 256
    print(f"L={L} F={F} T=2^{lg}: workspace {need.value/1e6:.0f} MB, slices/level {nb}, cap {cap}, records total {cnt.sum().item():.3e} "
          f"(of {R*S*8*L:.3e} corner terms); per level mean {[int(x) for x in cnt.mean(1).tolist()]}; max fill {cnt.max().item()/cap:.2f}")
