"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/neurad_hip.h declares, and the ctypes prototype table matches the header (no compute calls)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "neurad_hip.h")


def header_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(?:int|const char\*)\s+(nrhip_\w+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as ge

    if not os.path.exists(ge.LIB):
        ge.build()
    return ctypes.CDLL(ge.LIB)


def test_header_declares_the_path():
    fns = header_functions()
    for needed in ["nrhip_hashgrid_fwd", "nrhip_hashgrid_bwd", "nrhip_field_fwd", "nrhip_render_fwd",
                   "nrhip_proposal_density_fwd", "nrhip_pdf_sample", "nrhip_proposal_sampler_fwd",
                   "nrhip_composite_fwd", "nrhip_render_weight_from_alpha", "nrhip_mlp_fwd", "nrhip_last_error"]:
        assert needed in fns


def test_library_exports_every_declared_symbol(lib):
    for fn in header_functions():
        assert hasattr(lib, fn), f"{fn} declared in include/neurad_hip.h but not exported"


def test_ctypes_prototypes_match_header():
    from neurad_studio_amd import _lib

    declared = set(header_functions()) - {"nrhip_last_error"}
    assert declared == set(_lib.PROTOTYPES), (declared ^ set(_lib.PROTOTYPES))
    src = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    for name, argtypes in _lib.PROTOTYPES.items():
        m = re.search(name + r"\s*\((.*?)\)\s*;", src, flags=re.S)
        assert m, name
        args = [a for a in m.group(1).split(",") if a.strip() and a.strip() != "void"]
        assert len(args) == len(argtypes), f"{name}: header has {len(args)} args, ctypes table {len(argtypes)}"


def test_struct_layouts_match_header_sizes(lib):
    from neurad_studio_amd import _lib

    assert ctypes.sizeof(_lib.Grid) == 16 + 4 * 32
    assert ctypes.sizeof(_lib.Mlp) == 16 + 8 * 8 * 2
    assert ctypes.sizeof(_lib.Rays) == 72
    assert ctypes.sizeof(_lib.RgbDecoder) == 24 + 16 + 6 * 64 + 2 * 32 + 32  # 5 int32 (+ pad), 2 + 48 + 4 pointers, 16 floats


def test_rgb_decoder_sizes_is_host_logic_and_matches_the_module_tree(lib):
    """nrhip_rgb_decoder_sizes runs without a GPU: the gradient buffer it describes has one slot per parameter element of the
    reference-shaped decoder, in module order (what ops_decoder.RgbDecoderFn slices it by)."""
    from neurad_studio_amd import _lib
    from neurad_studio_amd.model_components.cnns import _fused_decoder_args, make_rgb_decoder

    dec = make_rgb_decoder(48, 32, 3)
    params, states, _ = _fused_decoder_args(dec)
    d = _lib.RgbDecoder()
    d.n_patches, d.patch_h, d.patch_w, d.cin, d.training = 40, 32, 32, 48, 1
    one = 0x1000  # any non-null address: nothing is dereferenced
    d.conv_in_w = d.conv_in_b = d.up_w = d.up_b = d.out_w = d.out_b = one
    for i in range(8):
        d.conv_w[i] = d.conv_b[i] = d.bn_gamma[i] = d.bn_beta[i] = d.bn_running_mean[i] = d.bn_running_var[i] = one
    sizes = [ctypes.c_int64(0) for _ in range(3)]
    _lib.call("nrhip_rgb_decoder_sizes", ctypes.byref(d), *(ctypes.byref(v) for v in sizes))
    assert sizes[2].value == sum(p.numel() for p in params) == sum(p.numel() for p in dec.parameters())
    n_lo, n_hi = 40 * 32 * 32, 40 * 96 * 96
    assert sizes[0].value >= (9 * n_lo + 9 * n_hi) * 64 and sizes[1].value >= 3 * n_hi * 64
    d.cin = 65  # unsupported width: an error code and a message, no crash
    with pytest.raises(_lib.NeuradHipError):
        _lib.call("nrhip_rgb_decoder_sizes", ctypes.byref(d), *(ctypes.byref(v) for v in sizes))


def test_version_and_error_string(lib):
    lib.nrhip_version.restype = ctypes.c_int
    assert lib.nrhip_version() >= 100
    lib.nrhip_last_error.restype = ctypes.c_char_p
    # argument validation happens on the host before any launch: NULL descriptor -> INVALID_ARG + message
    lib.nrhip_hashgrid_fwd.restype = ctypes.c_int
    rc = lib.nrhip_hashgrid_fwd(None, None, None, ctypes.c_int64(0), None, None)
    assert rc == 1 and b"NULL" in lib.nrhip_last_error()


def test_product_path_refuses_cpu_tensors():
    import torch

    from neurad_studio_amd import _lib, ops

    spec = ops.GridSpec(4, 2, 8, 16, 128)
    with pytest.raises(_lib.NeuradHipError):
        ops.hashgrid_fwd(spec, torch.zeros(4 * 256, 2), torch.rand(5, 3))


def test_round5_entry_points_validate_on_the_host_and_refuse_cpu_tensors(lib):
    """nrhip_encode_bwd_rays / nrhip_actor_pairs_* / nrhip_ray_order_large: argument checks run on the host before any launch
    (no GPU needed), and their tensor-level wrappers have no CPU path"""
    import torch

    from neurad_studio_amd import _lib, ops

    spec = ops.GridSpec(4, 2, 8, 16, 128)
    z = torch.zeros
    with pytest.raises(_lib.NeuradHipError):  # CPU tensors: refused, not computed on the host
        ops.encode_bwd_rays(spec, z(4 * 256, 2), 1.0, z(3, 3), z(3, 3), z(3), z(3, 5), z(3, 5), z(15, 8))
    with pytest.raises(_lib.NeuradHipError):
        ops.actor_pairs(z((10, 8), dtype=torch.int32))
    with pytest.raises(_lib.NeuradHipError):
        ops.ray_order(z(20000, 3), z(20000, 3), 100.0)
    need = ctypes.c_int64(0)
    _lib.call("nrhip_ray_order_workspace", 70000, 0, ctypes.byref(need))
    assert need.value == (70000 + 4096) * 4
    _lib.call("nrhip_ray_order_workspace", 70000, 5, ctypes.byref(need))
    assert need.value == (70000 + 32768) * 4
    one = ctypes.c_void_p(0x1000)  # any non-null address: validation fails before anything is dereferenced
    with pytest.raises(_lib.NeuradHipError, match="workspace"):  # workspace too small
        _lib.call("nrhip_ray_order_large", one, one, 70000, 1.0, 100.0, 0, one, 16, one, None)
    with pytest.raises(_lib.NeuradHipError, match="key_bits"):
        _lib.call("nrhip_ray_order_large", one, one, 70000, 1.0, 100.0, 3, one, 1 << 30, one, None)
    g = _grid(4, 2, 8)
    r = _lib.Rays()
    r.n_rays, r.n_samples = 3, 5
    r.origins = r.directions = r.pixel_area = r.starts = r.ends = 0x1000
    with pytest.raises(_lib.NeuradHipError, match="bad argument"):  # NULL gradient buffers
        _lib.call("nrhip_encode_bwd_rays", ctypes.byref(g), one, 1.0, ctypes.byref(r), one, None, None, None)
    total = ctypes.c_int64(0)
    with pytest.raises(_lib.NeuradHipError):  # NULL hits with samples to look at
        _lib.call("nrhip_actor_pairs_count", None, 10, one, one, None)


def _grid(L, F, lg):
    from neurad_studio_amd import _lib

    g = _lib.Grid()
    g.num_levels, g.n_features, g.log2_table_size, g.param_dtype = L, F, lg, 0
    for i in range(L):
        g.scalings[i] = 16.0 * (i + 1)
    return g


def test_table_gradient_workspace_query_is_host_logic(lib):
    """nrhip_encode_bwd_binned_workspace needs no GPU: one record slot per corner term ({entry, F values}, 8 per sample and level) --
    at F = 1 per x-pair of corner terms ({entry | xm, 2 values}, 4 per sample and level; the scalings here stay below every
    slice length, so no pair can straddle two slices) -- + bookkeeping, flat beyond 2^23 samples (larger batches go through in rounds), 0 for grids with more than 2048 slices per
    level."""
    fn = lib.nrhip_encode_bwd_binned_workspace
    fn.restype = ctypes.c_int

    def need(L, F, lg, n):
        out = ctypes.c_int64(-1)
        assert fn(ctypes.byref(_grid(L, F, lg)), ctypes.c_int64(n), ctypes.byref(out)) == 0
        return out.value

    assert need(16, 2, 19, 0) == 0
    records = lambda L, F, n: n * L * (4 * 3 if F == 1 else 8 * (F + 1)) * 4  # noqa: E731
    for L, F, lg in [(16, 2, 19), (8, 4, 22), (6, 1, 20), (4, 4, 17)]:
        a, b = need(L, F, lg, 4096 * 128), need(L, F, lg, 2 * 4096 * 128)
        assert records(L, F, 4096 * 128) <= a <= records(L, F, 4096 * 128) + (64 << 20)
        assert a < b and need(L, F, lg, 1 << 23) == need(L, F, lg, 1 << 24) > need(L, F, lg, 1 << 22)
    assert need(8, 8, 24, 4096) == 0  # 2^24 entries x 8 features: 8192 slices per level -> atomic entry point
    bad = ctypes.c_int64(0)
    assert fn(ctypes.byref(_grid(0, 2, 19)), ctypes.c_int64(16), ctypes.byref(bad)) != 0  # invalid grid is an error


def test_mlp_backward_workspace_query(lib):
    from neurad_studio_amd import _lib

    fn = lib.nrhip_mlp_bwd_workspace
    fn.restype = ctypes.c_int

    def need(i, h, o, nl, n):
        m = _lib.Mlp()
        m.in_dim, m.hidden_dim, m.out_dim, m.num_layers = i, h, o, nl
        for k in range(nl):
            m.weight[k] = 1  # never dereferenced by the query
        out = ctypes.c_int64(-1)
        assert fn(ctypes.byref(m), ctypes.c_int64(n), ctypes.byref(out)) == 0
        return out.value

    n = 1000
    assert need(13, 24, 3, 4, n) == (n * 3 * 24 + 3) // 4 * 4           # generic shape: dZ of the hidden layers only
    assert need(32, 64, 33, 2, n) > n * 64                              # NeuRAD's geometry MLP: + weight-gradient partials
    assert need(48, 64, 32, 3, n) > n * 128


def test_host_side_argument_validation_of_the_batch_samplers_and_feature_head(lib):
    """argument checks that run on the host before any launch: C entry points return INVALID_ARG / UNSUPPORTED with a
    message, the Python mirrors refuse CPU batches (no fallback)"""
    import torch

    from neurad_studio_amd import _lib
    from neurad_studio_amd.data.pixel_samplers import (LidarPointSamplerConfig, ScaledPatchSamplerConfig, lidar_point_sample,
                                                       patch_sample)

    I32, I64 = ctypes.c_int32, ctypes.c_int64
    lib.nrhip_patch_sample.restype = ctypes.c_int
    # rgb patch (4 * 3 = 12) larger than the 8 x 8 image
    rc = lib.nrhip_patch_sample(None, None, I64(2), I32(1), I32(8), I32(8), I32(3), I32(4), I32(3), None, None, I32(0), None,
                                None, None, None)
    assert rc != 0 and b"exceeds the image" in lib.nrhip_last_error()
    rc = lib.nrhip_patch_sample(None, None, I64(0), I32(1), I32(8), I32(8), I32(3), I32(2), I32(1), None, None, I32(0), None,
                                None, None, None)
    assert rc == 0  # empty batch: nothing to do, no pointer is touched
    rc = lib.nrhip_patch_sample(None, None, I64(2), I32(1), I32(8), I32(8), I32(3), I32(2), I32(1), None, None, I32(7), None,
                                None, None, None)
    assert rc != 0 and b"image_dtype" in lib.nrhip_last_error()
    lib.nrhip_lidar_point_sample.restype = ctypes.c_int
    rc = lib.nrhip_lidar_point_sample(None, None, None, None, None, I32(3000), I32(4), I32(5), I64(16), None, None, None)
    assert rc != 0 and b"2048" in lib.nrhip_last_error()
    rc = lib.nrhip_lidar_point_sample(None, None, None, None, None, I32(3), I32(4), I32(5), I64(13), None, None, None)
    assert rc != 0 and b"exceeds" in lib.nrhip_last_error()  # 13 rays from 3 x 4 draws
    # feature head backward: only the 48 -> H -> H -> 32 shapes
    m = _lib.Mlp()
    m.in_dim, m.hidden_dim, m.out_dim, m.num_layers = 64, 64, 32, 3
    for k in range(3):
        m.weight[k] = 1
    pw = (ctypes.c_void_p * _lib.MAX_LAYERS)()
    lib.nrhip_field_feature_bwd.restype = ctypes.c_int
    rc = lib.nrhip_field_feature_bwd(ctypes.byref(m), None, None, None, None, I64(16), None, pw, pw, None, I64(0), None)
    assert rc != 0 and b"feature head" in lib.nrhip_last_error()
    # Python mirrors: CPU batches are refused, malformed draws are caught before the call
    with pytest.raises(RuntimeError, match="GPU"):
        ScaledPatchSamplerConfig(patch_size=2, patch_scale=1).setup(num_rays_per_batch=8).sample(
            {"image": torch.rand(1, 8, 8, 3), "image_idx": torch.tensor([0])})
    with pytest.raises(RuntimeError, match="GPU"):
        LidarPointSamplerConfig().setup(num_rays_per_batch=8).sample(
            {"lidar": torch.rand(20, 5), "lidar_idx": torch.tensor([0, 1]), "points_per_lidar": torch.tensor([12, 8])})
    with pytest.raises(ValueError):
        patch_sample(torch.rand(1, 8, 8, 3), 2, 1)  # neither uniforms nor centers
    with pytest.raises(ValueError):
        lidar_point_sample(torch.rand(20, 5), torch.tensor([12, 8]), 8, shuffle=torch.tensor([0, 1]),
                           draws=torch.rand(2, 3, dtype=torch.float64))  # draws must be [2, ceil(8 / 2)]
    with pytest.raises(NotImplementedError):
        LidarPointSamplerConfig().setup(num_rays_per_batch=8).sample({"lidar": [torch.rand(20, 5)], "lidar_idx": torch.tensor([0])})


def test_power_sampler_ordered_host_checks(lib):
    """nrhip_power_sampler_ordered validates on the host: empty batch is a no-op, bad sizes / NULL buffers are errors"""
    I32, I64, F32 = ctypes.c_int32, ctypes.c_int64, ctypes.c_float
    fn = lib.nrhip_power_sampler_ordered
    fn.restype = ctypes.c_int
    args = lambda r, s: (None, None, I64(r), I32(s), F32(-1.0), F32(0.1), None, F32(0.0), None, None, None, None, F32(1.0),  # noqa: E731
                         F32(100.0), I32(0), None, None)
    assert fn(*args(0, 8)) == 0
    assert fn(*args(-1, 8)) != 0 and b"bad argument" in lib.nrhip_last_error()
    assert fn(*args(4, 0)) != 0
    assert fn(*args(4, 8)) != 0 and b"NULL" in lib.nrhip_last_error()


def test_eval_layout_plan_is_host_logic(lib):
    """nrhip_eval_layout_plan needs no GPU: levels whose lattice, padded to 2^s per axis, fits the table get a shadow region
    (power-of-two multipliers: OR == XOR), the others keep the reference's primes; regions are back to back."""
    fn = lib.nrhip_eval_layout_plan
    fn.restype = ctypes.c_int
    g = _grid(16, 2, 19)
    for i in range(16):  # BASELINE config[1]: floor(16 * g^l), g = 64^(1/15)
        g.scalings[i] = float(int(16 * (64 ** (1 / 15)) ** i + 1e-9))
    lay = (ctypes.c_uint32 * 64)()
    rows = ctypes.c_int64(0)
    assert fn(ctypes.byref(g), lay, ctypes.byref(rows)) == 0
    acc = 0
    for l in range(16):
        my, mz, mask, row0 = lay[4 * l:4 * l + 4]
        assert row0 == acc
        if g.scalings[l] < 64:  # coordinates 0..ceil(scale) fit s bits with 3 s <= 19
            s = my.bit_length() - 1
            assert my == 1 << s and mz == 1 << (2 * s) and mask == (1 << (3 * s)) - 1 and (1 << s) > g.scalings[l]
            acc += 1 << (3 * s)
        else:
            assert (my, mz, mask) == (2654435761, 805459861, (1 << 19) - 1)
            acc += 1 << 19
    assert rows.value == acc < 16 << 19
    assert fn(ctypes.byref(g), None, ctypes.byref(rows)) != 0

