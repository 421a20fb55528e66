"""BASELINE config[2]/[3] shape: full NeuRAD hot path (proposal sampler 128->64->32 + default field + compositing)."""
import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from neurad_studio_amd.models.neurad import NeuRADHotPath, NeuRADHotPathConfig
from neurad_studio_amd.cameras.rays import RayBundle
R = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
torch.manual_seed(0)
m = NeuRADHotPath(NeuRADHotPathConfig(appearance_dim=0), static_scale=100.0).cuda().eval()
g = torch.Generator(device="cuda"); g.manual_seed(0)
o = torch.randn((R, 3), device="cuda", generator=g) * 5
d = torch.randn((R, 3), device="cuda", generator=g); d = d / d.norm(dim=-1, keepdim=True)
def bundle():
    return RayBundle(origins=o, directions=d, pixel_area=torch.full((R, 1), 2.7e-7, device="cuda"),
                     nears=torch.zeros((R, 1), device="cuda"), fars=torch.full((R, 1), 20000.0, device="cuda"))
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
with torch.no_grad():
    ms = timeit(lambda: m.get_nff_outputs(bundle()))
    pf = [m.proposal_fields[1]] * 2
    rb = bundle(); m._prepare_bundle(rb)
    ms_s = timeit(lambda: m.sampler.generate_fused(rb, pf, 20000.0))
    rs, _, _ = m.sampler.generate_fused(rb, pf, 20000.0)
    fr = rs.frustums
    ms_r = timeit(lambda: m.field.render(rb.origins, rb.directions, rb.pixel_area, fr.starts[..., 0], fr.ends[..., 0]))
print(f"R={R}: get_nff_outputs (eval, fused) {ms:.3f} ms -> {R/ms*1e3:.3e} rays/s, {R*224/ms*1e3:.3e} field-evals/s | sampler kernel path {ms_s:.3f} ms | render {ms_r:.3f} ms")
