

def test_decode_rgb_recognises_only_the_reference_decoder_shape_and_falls_back_to_the_modules_on_cpu():
    """decode_rgb dispatches to the HIP decoder for the module tree of models/neurad.py:201-216 only; anything else -- and
    any CPU tensor -- runs the torch modules (no device, no library call)."""
    import torch

    from neurad_studio_amd.model_components.cnns import _fused_decoder_args, decode_rgb, make_rgb_decoder

    dec = make_rgb_decoder(48, 32, 3)
    args = _fused_decoder_args(dec)
    assert args is not None and len(args[0]) == 38 and len(args[1]) == 8 and len(args[2]) == 8
    assert [tuple(p.shape) for p in args[0][:6]] == [(32, 48, 1, 1), (32,), (32, 32, 7, 7), (32,), (32,), (32,)]
    assert _fused_decoder_args(make_rgb_decoder(48, 16, 3)) is None          # other width
    assert _fused_decoder_args(make_rgb_decoder(48, 32, 2)) is None          # other upsampling factor
    assert _fused_decoder_args(torch.nn.Sequential(torch.nn.Conv2d(48, 3, 1))) is None
    no_bn = make_rgb_decoder(48, 32, 3)
    no_bn[2].main_branch[1] = torch.nn.Identity()
    assert _fused_decoder_args(no_bn) is None
    x = torch.randn(2 * 64, 48)
    dec.eval()
    with torch.no_grad():
        got = decode_rgb(dec, x, (8, 8))
        want = dec(x.view(2, 8, 8, 48).permute(0, 3, 1, 2)).permute(0, 2, 3, 1)
    assert got.shape == (2, 24, 24, 3) and torch.equal(got, want)
