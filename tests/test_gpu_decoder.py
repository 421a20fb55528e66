"""RGB CNN decoder kernels (csrc/decoder.hip) against torch on the same fp16-rounded operands.
Tolerances: the kernels take fp16 operands and accumulate in fp32 like the reference's mixed-precision trainer; against an
fp32 convolution of the SAME rounded operands what is left is the summation order and the final rounding to fp16:
|err| <= 2e-3 * max|ref| (an fp16 ulp at the output's scale is 1e-3)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _nhwc16(x):  # [B,C,H,W] fp32 -> NHWC fp16
    return x.permute(0, 2, 3, 1).contiguous().half()


@pytest.mark.parametrize("shape,rows", [((3, 32, 32), 1), ((3, 32, 32), 2), ((2, 96, 96), 4), ((2, 40, 50), 2), ((1, 7, 33), 4)])
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
