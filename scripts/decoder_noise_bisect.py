"""Where does the HIP RGB decoder's extra gradient noise come from?  (round-5 review, item 6 b: the early layers' weight gradients
are 1.3 x further from the reference's fp32 model than torch autocast's are.)

The decoder is run STAGE BY STAGE over the library's per-stage entry points (the same kernels nrhip_rgb_decoder_fwd / _bwd chain,
csrc/decoder.hip:1459-1566), every intermediate kept; then
  (1) each backward op is checked IN ISOLATION: its output against a torch fp32 evaluation of the same op on the same (fp16)
      inputs -- what the op itself adds beyond the rounding of its output;
  (2) SUBSTITUTION: the backward is re-run with one class of ops (BatchNorm backward / input-gradient convolution / weight-gradient
      convolution / residual add) replaced by that fp32 evaluation (rounded to fp16 where the chain carries fp16), and the error
      of every weight gradient against the reference's fp32 model is compared with torch autocast's.
usage: python scripts/decoder_noise_bisect.py [patches] [patch_size] [seed]   -> a table on stdout (+ a RATIOS line: HIP error /
loss-scaled autocast error per weight tensor, for averaging over seeds)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F

from neurad_studio_amd import ops_decoder as D
from neurad_studio_amd.model_components.cnns import _fused_decoder_args, decode_rgb, make_rgb_decoder

B = int(sys.argv[1]) if len(sys.argv) > 1 else 3
P = int(sys.argv[2]) if len(sys.argv) > 2 else 24
SEED = int(sys.argv[3]) if len(sys.argv) > 3 else 0
torch.manual_seed(SEED)
dev = "cuda"
dec = make_rgb_decoder(48, 32, 3).to(dev).train()
feats = torch.randn((B * P * P, 48), device=dev)
image = torch.rand((B, 3 * P, 3 * P, 3), device=dev)


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / (b.norm() + 1e-300))


def nchw(t):  # [B,H,W,32] -> [B,32,H,W] fp32
    return t.float().permute(0, 3, 1, 2).contiguous()


def nhwc16(t):
    return t.permute(0, 2, 3, 1).contiguous().half()


def loss_of(rgb):
    return F.mse_loss(rgb, image)


def run_torch(autocast, loss_scale=1.0):
    for p in dec.parameters():
        p.grad = None
    f = feats.clone().requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.float16, enabled=autocast):
        rgb = decode_rgb(dec, f, (P, P), fused=False).float()
    (loss_of(rgb) * loss_scale).backward()
    return {n: p.grad.clone() / loss_scale for n, p in dec.named_parameters()}


ref = run_torch(False)
auto = run_torch(True)
auto_scaled = {k: run_torch(True, 2.0 ** k) for k in (8, 16)}  # what the trainer's GradScaler does for the torch modules
params, states, bns = _fused_decoder_args(dec)
names = [n for n, _ in dec.named_parameters()]
pmap = {id(p): n for n, p in dec.named_parameters()}
pn = [pmap[id(p)] for p in params]  # names in the library's parameter order

for p in dec.parameters():
    p.grad = None
f = feats.clone().requires_grad_(True)
loss_of(decode_rgb(dec, f, (P, P), fused=True)).backward()
hip = {n: p.grad.clone() for n, p in dec.named_parameters()}


# ---- the staged chain ------------------------------------------------------------------------------------------------------
def staged(substitute=()):
    """-> ({param name: grad}, {op label: rel. error of the op's output against its fp32 evaluation on the same inputs})"""
    w0, b0 = params[0], params[1]
    blocks = [params[2 + 8 * k:10 + 8 * k] for k in range(4)]
    wu, bu, wo, bo = params[34:38]
    packed = D.conv7x7_pack_many([blk[j] for blk in blocks for j in (0, 4)])
    wup = D.upsample_pack(wu)
    sv = {}
    h0 = D.conv1x1_in_fwd(feats, w0, b0).view(B, P, P, 32)
    x = h0
    up_in = None
    for k in range(4):
        if k == 2:
            up_in = x
            x = D.upsample_fwd(x, wup, bu)
        wa, ba, g1, be1, wb, bb, g2, be2 = blocks[k]
        n = x.numel() // 32
        c1, st = D.conv7x7(x, packed[2 * k, 0], ba, stats=True)
        cf1 = D.bn_finalize(st, n, g1, be1, 1e-5, 0.1, None, None)
        u1 = D.bn_act(c1, cf1)
        c2, st = D.conv7x7(u1, packed[2 * k + 1, 0], bb, stats=True)
        cf2 = D.bn_finalize(st, n, g2, be2, 1e-5, 0.1, None, None)
        out = D.bn_act(c2, cf2, skip=x)
        sv[k] = (x, c1, cf1, u1, c2, cf2, out)
        x = out
    rgb = D.rgb_fwd(x, wo, bo)
    grad_rgb = torch.autograd.grad(loss_of(rgb.requires_grad_(True)), rgb)[0].contiguous()
    gs = D.grad_scale(grad_rgb)
    S, inv = float(gs[0]), float(gs[1])
    g = {n: torch.zeros_like(p) for n, p in zip(pn, params)}
    iso = {}

    def bn_bwd(dout, act, c, gamma, coef, gname, bname, label):
        got = D.bn_bwd(dout, act, c, gamma, coef, g[gname], g[bname], gs)
        mean, rstd = coef[2], coef[3]
        dy = dout.float() * (act > 0)
        xh = (c.float() - mean) * rstd
        n = c.numel() // 32
        dg, db = (dy * xh).sum((0, 1, 2)), dy.sum((0, 1, 2))
        want = (gamma * rstd / n) * (n * dy - db - xh * dg)
        iso[label] = rel(got, want)
        if "bn" in substitute:
            g[gname].copy_(dg * inv), g[bname].copy_(db * inv)
            return want.half()
        return got

    def dgrad(dc, w, pk, label):
        got, _ = D.conv7x7(dc, pk)
        want = F.conv_transpose2d(nchw(dc), w.half().float(), padding=3)
        iso[label] = rel(nchw(got), want)
        return nhwc16(want) if "dgrad" in substitute else got

    def wgrad(xin, dc, wname, bname, label):
        D.conv7x7_wgrad(xin, dc, g[wname], g[bname], gs)
        want = torch.nn.grad.conv2d_weight(nchw(xin), (32, 32, 7, 7), nchw(dc), padding=3) * inv
        iso[label] = rel(g[wname], want)
        if "wgrad" in substitute:
            g[wname].copy_(want), g[bname].copy_(dc.float().sum((0, 1, 2)) * inv)

    dcur = D.rgb_bwd(x, rgb.detach(), grad_rgb, wo, g[pn[36]], g[pn[37]], gs)
    for k in (3, 2, 1, 0):
        xin, c1, cf1, u1, c2, cf2, out = sv[k]
        wa, ba, g1, be1, wb, bb, g2, be2 = blocks[k]
        na = pn[2 + 8 * k:10 + 8 * k]
        dc2 = bn_bwd(dcur, out, c2, g2, cf2, na[6], na[7], f"block{k}.bn2_bwd")
        du1 = dgrad(dc2, wb, packed[2 * k + 1, 1], f"block{k}.conv2_dgrad")
        wgrad(u1, dc2, na[4], na[5], f"block{k}.conv2_wgrad")
        dc1 = bn_bwd(du1, u1, c1, g1, cf1, na[2], na[3], f"block{k}.bn1_bwd")
        dx = dgrad(dc1, wa, packed[2 * k, 1], f"block{k}.conv1_dgrad")
        wgrad(xin, dc1, na[0], na[1], f"block{k}.conv1_wgrad")
        got = D.add_masked(dx, dcur, out)
        want = dx.float() + dcur.float() * (out > 0)
        iso[f"block{k}.residual_add"] = rel(got, want)
        dcur = want.half() if "add" in substitute else got
        if k == 2:
            dcur = D.upsample_bwd(up_in, dcur, wup, g[pn[34]], g[pn[35]], gs)
    D.conv1x1_in_bwd(feats, h0.view(-1, 32), dcur.view(-1, 32), w0, g[pn[0]], g[pn[1]], gs)
    dh = dcur.view(-1, 32).float() * (h0.view(-1, 32) > 0)
    iso["conv_in.wgrad"] = rel(g[pn[0]].view(32, -1), dh.t() @ feats * inv)
    return g, iso


conv_w = [n for n in names if n.endswith(("main_branch.0.weight", "main_branch.3.weight"))]
other_w = [n for n in names if n.endswith("weight") and n not in conv_w and "main_branch" not in n]  # 1x1 in, upsample, rgb head
g0, iso = staged()
print(f"# {B} patches of {P} x {P}; rel-L2 of d loss / d (7x7 weights) against the reference's fp32 modules")
print("staged chain == monolithic call:", max(rel(g0[n], hip[n]) for n in conv_w))
print("\n(1) each backward op against its fp32 evaluation on the SAME inputs (rel-L2 of the op's output):")
for k, v in iso.items():
    print(f"   {k:26s} {v:.2e}")
runs = {"autocast x 2^8": auto_scaled[8], "autocast x 2^16": auto_scaled[16], "HIP decoder": g0}
for sub in (("bn", "dgrad", "wgrad", "add"),):
    runs["HIP, fp32 " + "+".join(sub)] = staged(sub)[0]
print("\n(2) weight-gradient error per 7x7 convolution (first = nearest the input), and the ratio to torch autocast's:")
print(f"   {'layer':24s} {'autocast':>9s} " + " ".join(f"{k[:24]:>24s}" for k in runs))
for n in other_w[:1] + conv_w + other_w[1:]:
    a = rel(auto[n], ref[n])
    print(f"   {n:24s} {a:9.3f} " + " ".join(f"{rel(r[n], ref[n]):15.3f} ({rel(r[n], ref[n]) / a:4.2f}x)" for r in runs.values()))
print("RATIOS", SEED, " ".join(f"{rel(g0[n], ref[n]) / rel(auto_scaled[16][n], ref[n]):.3f}" for n in other_w[:1] + conv_w + other_w[1:2]))
