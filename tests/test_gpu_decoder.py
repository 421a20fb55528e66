"""RGB CNN decoder kernels (csrc/decoder.hip) against torch on the same fp16-rounded operands.
Tolerances: the kernels take fp16 operands and accumulate in fp32 like the reference's mixed-precision trainer; against an
fp32 convolution of the SAME rounded operands what is left is the summation order and the final rounding to fp16:
|err| <= 2e-3 * max|ref| (an fp16 ulp at the output's scale is 1e-3)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _nhwc16(x):  # [B,C,H,W] fp32 -> NHWC fp16
    return x.permute(0, 2, 3, 1).contiguous().half()


@pytest.mark.parametrize("shape,rows", [((3, 32, 32), 1), ((3, 32, 32), 2), ((2, 96, 96), 4), ((2, 40, 50), 2), ((1, 7, 33), 4),
                                        ((30, 96, 96), 4), ((29, 90, 70), 4)])  # the last two: >= 2 tiles per CU -> persistent kernel
def test_conv7x7_forward_and_stats_match_torch(shape, rows):
    from neurad_studio_amd import ops_decoder as D

    B, H, W = shape
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn((B, 32, H, W), device="cuda", generator=g)
    w = torch.randn((32, 32, 7, 7), device="cuda", generator=g) * 0.05
    bias = torch.randn((32,), device="cuda", generator=g)
    xh = _nhwc16(x)
    ref = torch.nn.functional.conv2d(xh.float().permute(0, 3, 1, 2), w.half().float(), bias, padding=3)
    out, part = D.conv7x7(xh, D.conv7x7_pack(w, 0), bias, stats=True, rows_per_wave=rows)
    got = out.float().permute(0, 3, 1, 2)
    assert (got - ref).abs().max() <= 2e-3 * ref.abs().max()
    # BatchNorm's batch statistics are taken from the rounded outputs
    s = part.double().sum(0)
    o = out.double().reshape(-1, 32)
    assert torch.allclose(s[:32], o.sum(0), rtol=1e-5, atol=1e-3)
    assert torch.allclose(s[32:], o.square().sum(0), rtol=1e-5, atol=1e-3)


def test_conv7x7_on_mode1_weights_is_the_input_gradient():
    from neurad_studio_amd import ops_decoder as D

    g = torch.Generator(device="cuda").manual_seed(4)
    x = torch.randn((2, 32, 32, 32), device="cuda", generator=g, requires_grad=True)
    w = (torch.randn((32, 32, 7, 7), device="cuda", generator=g) * 0.05).half().float()
    gy = torch.randn((2, 32, 32, 32), device="cuda", generator=g)
    gyh = _nhwc16(gy)
    torch.nn.functional.conv2d(x, w, None, padding=3).backward(gyh.float().permute(0, 3, 1, 2))
    got, _ = D.conv7x7(gyh, D.conv7x7_pack(w, 1))
    ref = x.grad
    assert (got.float().permute(0, 3, 1, 2) - ref).abs().max() <= 2e-3 * ref.abs().max()


@pytest.mark.parametrize("shape", [(3, 32, 32), (2, 96, 96), (2, 20, 50), (1, 5, 17)])
def test_conv7x7_weight_and_bias_gradient_match_torch(shape):
    from neurad_studio_amd import ops_decoder as D

    B, H, W = shape
    g = torch.Generator(device="cuda").manual_seed(5)
    xh = _nhwc16(torch.randn((B, 32, H, W), device="cuda", generator=g))
    gh = _nhwc16(torch.randn((B, 32, H, W), device="cuda", generator=g))
    w = torch.zeros((32, 32, 7, 7), device="cuda", requires_grad=True)
    bias = torch.zeros((32,), device="cuda", requires_grad=True)
    torch.nn.functional.conv2d(xh.float().permute(0, 3, 1, 2), w, bias, padding=3).backward(gh.float().permute(0, 3, 1, 2))
    gw = torch.full((32, 32, 7, 7), 0.5, device="cuda")  # accumulated into
    gb = torch.full((32,), -1.0, device="cuda")
    D.conv7x7_wgrad(xh, gh, gw, gb)
    # exact products of fp16 operands, fp32 accumulation on both sides: only the summation order differs
    assert (gw - 0.5 - w.grad).abs().max() <= 2e-5 * w.grad.abs().max() + 1e-4
    assert (gb + 1.0 - bias.grad).abs().max() <= 2e-5 * bias.grad.abs().max() + 1e-4


def _h(t):  # the value an fp16 operand carries
    return t.half().float()


def test_first_and_last_layer_and_upsampling_match_torch_on_the_rounded_operands():
    from neurad_studio_amd import ops_decoder as D

    g = torch.Generator(device="cuda").manual_seed(6)
    n, cin = 2 * 20 * 20, 48
    f = torch.randn((n, cin), device="cuda", generator=g)
    w0 = torch.randn((32, cin, 1, 1), device="cuda", generator=g) * 0.2
    b0 = torch.randn((32,), device="cuda", generator=g) * 0.1
    # Conv2d(48, 32, 1) + ReLU and its backward
    fr, wr, br = _h(f).requires_grad_(), _h(w0).reshape(32, cin).requires_grad_(), b0.clone().requires_grad_()
    ref = torch.relu(fr @ wr.t() + br)
    h0 = D.conv1x1_in_fwd(f, w0, b0)
    assert (h0.float() - ref).abs().max() <= 2e-3 * ref.abs().max()
    dh = torch.randn((n, 32), device="cuda", generator=g).half()
    (h0.float().detach() > 0).float()  # mask comes from the rounded activation on both sides
    (torch.relu(fr @ wr.t() + br) * 1.0).backward(dh.float() * (h0.float() > 0) / (ref.detach() > 0).clamp(min=1))
    gw, gb = torch.zeros((32, cin), device="cuda"), torch.zeros((32,), device="cuda")
    gf = D.conv1x1_in_bwd(f, h0, dh, w0.reshape(32, cin), gw, gb)
    for got, want in ((gf, fr.grad), (gw, wr.grad), (gb, br.grad)):
        assert (got - want).abs().max() <= 1e-4 * want.abs().max() + 1e-5
    # ConvTranspose2d(32, 32, 3, stride 3)
    x = torch.randn((2, 10, 12, 32), device="cuda", generator=g).half()
    wu = torch.randn((32, 32, 3, 3), device="cuda", generator=g) * 0.1
    bu = torch.randn((32,), device="cuda", generator=g) * 0.1
    xr, wur, bur = x.float().permute(0, 3, 1, 2).requires_grad_(), _h(wu).requires_grad_(), bu.clone().requires_grad_()
    ref = torch.nn.functional.conv_transpose2d(xr, wur, bur, stride=3)
    wup = D.upsample_pack(wu)
    up = D.upsample_fwd(x, wup, bu)
    assert (up.float().permute(0, 3, 1, 2) - ref).abs().max() <= 2e-3 * ref.abs().max()
    dup = torch.randn((2, 30, 36, 32), device="cuda", generator=g).half()
    ref.backward(dup.float().permute(0, 3, 1, 2))
    gwu, gbu = torch.zeros_like(wu), torch.zeros_like(bu)
    gx = D.upsample_bwd(x, dup, wup, gwu, gbu)
    assert (gx.float().permute(0, 3, 1, 2) - xr.grad).abs().max() <= 2e-3 * xr.grad.abs().max()
    assert (gwu - wur.grad).abs().max() <= 1e-4 * wur.grad.abs().max()
    assert (gbu - bur.grad).abs().max() <= 1e-4 * bur.grad.abs().max()
    # Conv2d(32, 3, 1) + Sigmoid
    wo = torch.randn((3, 32, 1, 1), device="cuda", generator=g) * 0.3
    bo = torch.randn((3,), device="cuda", generator=g) * 0.1
    hr, wor, bor = x.float().requires_grad_(), _h(wo).reshape(3, 32).requires_grad_(), bo.clone().requires_grad_()
    ref = torch.sigmoid(hr @ wor.t() + bor)
    rgb = D.rgb_fwd(x, wo, bo)
    assert (rgb - ref).abs().max() <= 1e-5
    drgb = torch.randn(ref.shape, device="cuda", generator=g)
    ref.backward(drgb)
    gwo, gbo = torch.zeros((3, 32), device="cuda"), torch.zeros((3,), device="cuda")
    gh = D.rgb_bwd(x, rgb, drgb, wo, gwo, gbo)
    assert (gh.float() - hr.grad).abs().max() <= 2e-3 * hr.grad.abs().max()
    assert (gwo - wor.grad).abs().max() <= 1e-4 * wor.grad.abs().max()
    assert (gbo - bor.grad).abs().max() <= 1e-4 * bor.grad.abs().max()


def test_batch_norm_relu_forward_backward_and_running_statistics_match_torch():
    from neurad_studio_amd import ops_decoder as D

    g = torch.Generator(device="cuda").manual_seed(7)
    B, H, W = 3, 32, 32
    x = torch.randn((B, H, W, 32), device="cuda", generator=g).half()
    skip = torch.randn((B, H, W, 32), device="cuda", generator=g).half()
    w = torch.randn((32, 32, 7, 7), device="cuda", generator=g) * 0.05
    gamma = torch.rand((32,), device="cuda", generator=g) + 0.5
    beta = torch.randn((32,), device="cuda", generator=g) * 0.2
    c, part = D.conv7x7(x, D.conv7x7_pack(w, 0), None, stats=True)
    rm, rv = torch.zeros(32, device="cuda"), torch.ones(32, device="cuda")
    coef = D.bn_finalize(part, B * H * W, gamma, beta, 1e-5, 0.1, rm, rv)
    out = D.bn_act(c, coef, skip)
    cr = c.float().permute(0, 3, 1, 2).requires_grad_()
    gr, br = gamma.clone().requires_grad_(), beta.clone().requires_grad_()
    rm_t, rv_t = torch.zeros(32, device="cuda"), torch.ones(32, device="cuda")
    bn = torch.nn.functional.batch_norm(cr, rm_t, rv_t, gr, br, training=True, momentum=0.1, eps=1e-5)
    ref = torch.relu(bn + skip.float().permute(0, 3, 1, 2))
    assert (out.float().permute(0, 3, 1, 2) - ref).abs().max() <= 4e-3 * ref.abs().max()
    assert torch.allclose(rm, rm_t, atol=1e-5) and torch.allclose(rv, rv_t, rtol=1e-4)
    d = torch.randn((B, H, W, 32), device="cuda", generator=g).half()
    # the mask of the product path is the ROUNDED output's; give torch the same one
    mask = (out.float() > 0).permute(0, 3, 1, 2)
    bn.backward(d.float().permute(0, 3, 1, 2) * mask)
    gg, gb = torch.zeros(32, device="cuda"), torch.zeros(32, device="cuda")
    dc = D.bn_bwd(d, out, c, gamma, coef, gg, gb)
    assert (dc.float().permute(0, 3, 1, 2) - cr.grad).abs().max() <= 3e-3 * cr.grad.abs().max()
    assert (gg - gr.grad).abs().max() <= 1e-3 * gr.grad.abs().max()
    assert (gb - br.grad).abs().max() <= 1e-3 * br.grad.abs().max()
    got = D.add_masked(dc, d, out)
    assert (got.float() - (dc.float() + d.float() * (out.float() > 0))).abs().max() <= 2e-3 * dc.float().abs().max()


def test_decoder_end_to_end_against_the_torch_modules():
    """decode_rgb on the HIP kernels vs the same modules in fp32 torch: outputs, every parameter gradient, the feature
    gradient and BatchNorm's running statistics.  The yardstick for 'fp16 operands, fp32 accumulation' is the reference
    trainer's own path, torch autocast(fp16) on the same modules: the HIP path may not be further from fp32 than 1.5 x that
    (+ a floor)."""
    import copy

    from neurad_studio_amd.model_components.cnns import decode_rgb, make_rgb_decoder

    torch.manual_seed(8)
    dec = make_rgb_decoder(48, 32, 3).cuda().train()
    with torch.no_grad():
        for m in dec.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.uniform_(0.5, 1.5)
                m.bias.normal_(0, 0.2)
    B = 3
    f = torch.randn((B * 1024, 48), device="cuda")
    image = torch.rand((B, 96, 96, 3), device="cuda")

    def run(mode):
        d = copy.deepcopy(dec)
        x = f.clone().requires_grad_()
        if mode == "hip":
            rgb = decode_rgb(d, x, (32, 32))
        elif mode == "fp32":
            rgb = decode_rgb(d, x, (32, 32), fused=False)
        else:
            with torch.autocast("cuda", dtype=torch.float16):
                rgb = decode_rgb(d, x, (32, 32), fused=False)
        torch.nn.functional.mse_loss(rgb.float(), image).backward()
        return rgb.detach().float(), x.grad, {n: p.grad for n, p in d.named_parameters()}, dict(d.named_buffers())

    rgb_h, gx_h, gp_h, buf_h = run("hip")
    rgb_r, gx_r, gp_r, buf_r = run("fp32")
    rgb_a, gx_a, gp_a, _ = run("autocast")

    def rel(a, b):
        return float((a - b).norm() / (b.norm() + 1e-20))

    assert (rgb_h - rgb_r).abs().max() <= max(1.5 * float((rgb_a - rgb_r).abs().max()), 2e-3)
    assert rel(gx_h, gx_r) <= max(1.5 * rel(gx_a, gx_r), 5e-3)
    for n in gp_r:
        if n.endswith("main_branch.0.bias") or n.endswith("main_branch.3.bias"):
            continue  # a bias in front of BatchNorm has a zero gradient: only rounding noise on both sides
        assert rel(gp_h[n], gp_r[n]) <= max(1.5 * rel(gp_a[n], gp_r[n]), 5e-3), (n, rel(gp_h[n], gp_r[n]), rel(gp_a[n], gp_r[n]))
    for n in buf_r:
        if n.endswith("num_batches_tracked"):
            assert int(buf_h[n]) == int(buf_r[n]) == 1
        else:
            assert torch.allclose(buf_h[n], buf_r[n], rtol=2e-3, atol=2e-4), n
    # eval mode: running statistics
    d = copy.deepcopy(dec).eval()
    with torch.no_grad():
        a = decode_rgb(d, f, (32, 32))
        b = decode_rgb(d, f, (32, 32), fused=False)
    assert (a - b).abs().max() <= 3e-3


def test_decoder_backward_is_scale_free():
    """an upstream gradient of order 1e-9 (a mean over a million pixels, no loss scale) must give the same parameter
    gradients, relative to their size, as one of order 1: the backward's working scale keeps fp16 away from its subnormals.
    (Not bit-equal: 1e-9 is no power of two, so the scaled fp16 values round differently -- 4e-3 is a few fp16 ulps.)"""
    import copy

    from neurad_studio_amd.model_components.cnns import decode_rgb, make_rgb_decoder

    torch.manual_seed(9)
    dec = make_rgb_decoder(48, 32, 3).cuda().train()
    f = torch.randn((2 * 1024, 48), device="cuda")
    w = torch.randn((2, 96, 96, 3), device="cuda")
    grads = []
    for k in (1.0, 1e-9):
        d = copy.deepcopy(dec)
        x = f.clone().requires_grad_()
        (decode_rgb(d, x, (32, 32)) * w).sum().mul(k).backward()
        grads.append((x.grad / k, {n: p.grad / k for n, p in d.named_parameters()}))
    (gx1, gp1), (gx2, gp2) = grads
    assert float((gx1 - gx2).norm() / gx1.norm()) <= 4e-3
    for n in gp1:
        if n.endswith("main_branch.0.bias") or n.endswith("main_branch.3.bias"):
            continue
        assert float((gp1[n] - gp2[n]).norm() / (gp1[n].norm() + 1e-30)) <= 4e-3, n


def test_eval_mode_decoder_on_a_wide_chunk_matches_the_torch_modules():
    """eval renders image rows, not 32 x 32 patches (models/neurad.py:623-675 -> decode_features with the chunk's shape): a
    6 x 200 chunk through the HIP decoder with BatchNorm on its running statistics vs the torch modules in fp32."""
    from neurad_studio_amd.model_components.cnns import decode_rgb, make_rgb_decoder

    torch.manual_seed(11)
    dec = make_rgb_decoder(48, 32, 3).cuda()
    with torch.no_grad():
        for m in dec.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.3)
                m.running_var.uniform_(0.5, 1.5)
                m.weight.uniform_(0.5, 1.5)
    dec.eval()
    f = torch.randn((2 * 6 * 200, 48), device="cuda")
    with torch.no_grad():
        a = decode_rgb(dec, f, (6, 200))
        b = decode_rgb(dec, f, (6, 200), fused=False)
    assert a.shape == (2, 18, 600, 3) and (a - b).abs().max() <= 3e-3


def test_decoder_backward_in_eval_mode_matches_the_torch_modules():
    """eval mode (running statistics): BatchNorm is an affine map with constants, its backward A g alone.  Gradients of the
    features and of every parameter against the same modules in fp32 torch, torch autocast(fp16) as the yardstick."""
    import copy

    from neurad_studio_amd.model_components.cnns import decode_rgb, make_rgb_decoder

    torch.manual_seed(9)
    dec = make_rgb_decoder(48, 32, 3).cuda()
    with torch.no_grad():
        for m in dec.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.uniform_(0.5, 1.5)
                m.bias.normal_(0, 0.2)
                m.running_mean.normal_(0, 0.3)
                m.running_var.uniform_(0.5, 2.0)
    dec.eval()
    f = torch.randn((2 * 1024, 48), device="cuda")
    image = torch.rand((2, 96, 96, 3), device="cuda")

    def run(mode):
        d = copy.deepcopy(dec)
        x = f.clone().requires_grad_()
        if mode == "hip":
            rgb = decode_rgb(d, x, (32, 32))
        elif mode == "fp32":
            rgb = decode_rgb(d, x, (32, 32), fused=False)
        else:
            with torch.autocast("cuda", dtype=torch.float16):
                rgb = decode_rgb(d, x, (32, 32), fused=False)
        torch.nn.functional.mse_loss(rgb.float(), image).backward()
        return rgb.detach().float(), x.grad, {n: p.grad for n, p in d.named_parameters()}, dict(d.named_buffers())

    rgb_h, gx_h, gp_h, buf_h = run("hip")
    rgb_r, gx_r, gp_r, buf_r = run("fp32")
    _, gx_a, gp_a, _ = run("autocast")

    def rel(a, b):
        return float((a - b).norm() / (b.norm() + 1e-20))

    assert rel(rgb_h, rgb_r) < 3e-3
    assert rel(gx_h, gx_r) <= max(1.5 * rel(gx_a, gx_r), 5e-3)
    for n in gp_r:  # (in eval mode the convolution biases in front of BatchNorm DO have a gradient)
        assert rel(gp_h[n], gp_r[n]) <= max(1.5 * rel(gp_a[n], gp_r[n]), 5e-3), (n, rel(gp_h[n], gp_r[n]), rel(gp_a[n], gp_r[n]))
    for n in buf_r:  # nothing is tracked in eval mode
        assert torch.equal(buf_h[n], buf_r[n]), n
