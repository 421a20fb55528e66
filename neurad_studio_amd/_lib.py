"""ctypes binding of libneurad_hip.so (the C ABI declared in include/neurad_hip.h).

The product path has NO fallback: if the HIP library is missing or a symbol cannot be resolved,
loading raises.  Mirrors the behaviour of the reference's tcnn import gate
(nerfstudio/utils/external.py:18-58), except that nothing is lazily swallowed.
"""
from __future__ import annotations

import ctypes as C
import os

MAX_LEVELS = 32
MAX_LAYERS = 8

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NEURAD_HIP_LIB", os.path.join(_HERE, "lib", "libneurad_hip.so"))

c_float_p = C.POINTER(C.c_float)


class Grid(C.Structure):
    _fields_ = [("num_levels", C.c_int32), ("n_features", C.c_int32), ("log2_table_size", C.c_int32),
                ("param_dtype", C.c_int32), ("scalings", C.c_float * MAX_LEVELS)]


class Mlp(C.Structure):
    _fields_ = [("in_dim", C.c_int32), ("hidden_dim", C.c_int32), ("out_dim", C.c_int32), ("num_layers", C.c_int32),
                ("weight", C.c_void_p * MAX_LAYERS), ("bias", C.c_void_p * MAX_LAYERS)]


class Rays(C.Structure):
    _fields_ = [("n_rays", C.c_int64), ("n_samples", C.c_int32), ("origins", C.c_void_p),
                ("directions", C.c_void_p), ("pixel_area", C.c_void_p), ("starts", C.c_void_p),
                ("ends", C.c_void_p), ("sample_stride", C.c_int32), ("order", C.c_void_p)]


class Field(C.Structure):
    _fields_ = [("grid", Grid), ("table", C.c_void_p), ("static_scale", C.c_float), ("geo", Mlp), ("feat", Mlp),
                ("use_sdf", C.c_int32), ("beta", C.c_float), ("eval_table", C.c_void_p),
                ("eval_layout", C.POINTER(C.c_uint32))]


class Proposal(C.Structure):
    _fields_ = [("grid", Grid), ("table", C.c_void_p), ("static_scale", C.c_float), ("decoder_weight", C.c_void_p)]


class SamplerCfg(C.Structure):
    _fields_ = [("n_rounds", C.c_int32), ("n_samples", C.c_int32 * 3), ("lam", C.c_float), ("scaling", C.c_float),
                ("histogram_padding", C.c_float), ("sky_distance", C.c_float)]


class Actors(C.Structure):
    _fields_ = [("n_actors", C.c_int32), ("n_times", C.c_int32), ("timestamps", C.c_void_p), ("positions", C.c_void_p),
                ("rotations_6d", C.c_void_p), ("present", C.c_void_p), ("bounds", C.c_void_p), ("grid", Grid),
                ("tables", C.c_void_p), ("actor_scale", C.c_float), ("max_candidates", C.c_int32)]


class ActorEdit(C.Structure):  # nrhip_actor_edit
    _fields_ = [("lateral", C.c_float), ("longitudinal", C.c_float), ("height", C.c_float), ("rotation", C.c_float),
                ("index", C.c_int32)]


class RgbDecoder(C.Structure):
    _fields_ = [("n_patches", C.c_int32), ("patch_h", C.c_int32), ("patch_w", C.c_int32), ("cin", C.c_int32),
                ("training", C.c_int32), ("conv_in_w", C.c_void_p), ("conv_in_b", C.c_void_p),
                ("conv_w", C.c_void_p * 8), ("conv_b", C.c_void_p * 8), ("bn_gamma", C.c_void_p * 8),
                ("bn_beta", C.c_void_p * 8), ("bn_running_mean", C.c_void_p * 8), ("bn_running_var", C.c_void_p * 8),
                ("bn_eps", C.c_float * 8), ("bn_momentum", C.c_float * 8), ("up_w", C.c_void_p), ("up_b", C.c_void_p),
                ("out_w", C.c_void_p), ("out_b", C.c_void_p)]


class AdamTensor(C.Structure):  # nrhip_adam_tensor
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p),
                ("image_fp16", C.c_void_p), ("n", C.c_int64), ("step", C.c_int64), ("grad_dtype", C.c_int32),
                ("reserved", C.c_int32)]


class AdamTensorDev(C.Structure):  # nrhip_adam_tensor_dev
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p),
                ("image_fp16", C.c_void_p), ("n", C.c_int64), ("step", C.c_void_p), ("grad_dtype", C.c_int32),
                ("reserved", C.c_int32)]


class CheckTensor(C.Structure):  # nrhip_check_tensor
    _fields_ = [("data", C.c_void_p), ("n", C.c_int64), ("dtype", C.c_int32), ("reserved", C.c_int32)]


class OccGrid(C.Structure):
    _fields_ = [("aabb", C.c_float * 6), ("resolution", C.c_int32), ("binaries", C.c_void_p)]


class CameraTable(C.Structure):
    _fields_ = [("camera_to_worlds", C.c_void_p), ("fx", C.c_void_p), ("fy", C.c_void_p), ("cx", C.c_void_p),
                ("cy", C.c_void_p), ("times", C.c_void_p), ("rolling_shutter", C.c_int32),
                ("rolling_shutter_time", C.c_void_p), ("time_to_center_pixel", C.c_void_p), ("velocities", C.c_void_p),
                ("shutter_extent", C.c_void_p)]


class LidarTable(C.Structure):
    _fields_ = [("lidar_to_worlds", C.c_void_p), ("times", C.c_void_p), ("velocities", C.c_void_p),
                ("horizontal_beam_divergence", C.c_void_p), ("vertical_beam_divergence", C.c_void_p),
                ("assume_ego_compensated", C.c_int32), ("valid_lidar_distance_threshold", C.c_float)]


MAX_SAMPLE_CONTAINMENTS = 8  # NRHIP_MAX_SAMPLE_CONTAINMENTS
GRAD_ROWS_PER_BLOCK = 256   # NRHIP_GRAD_ROWS_PER_BLOCK
P, I32, I64, F32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float

# name -> argtypes : exactly the prototypes of include/neurad_hip.h (tests/test_abi.py cross-checks this
# table against the header so the two cannot drift)
PROTOTYPES = {
    "nrhip_version": [],
    "nrhip_device_info": [C.POINTER(I32), C.POINTER(I32), C.POINTER(I64)],
    "nrhip_eval_layout_plan": [C.POINTER(Grid), C.POINTER(C.c_uint32), C.POINTER(I64)],
    "nrhip_eval_layout_build": [C.POINTER(Grid), P, C.POINTER(C.c_uint32), P, P],
    "nrhip_conv7x7_pack": [P, I32, P, P],
    "nrhip_conv7x7_pack_many": [P, I32, P, P],
    "nrhip_conv7x7_tiles": [I32, I32, I32, C.POINTER(I32)],
    "nrhip_conv7x7": [P, P, P, P, P, I32, I32, I32, I32, P],
    "nrhip_conv7x7_wgrad_workspace": [I32, I32, I32, C.POINTER(I64)],
    "nrhip_conv7x7_wgrad": [P, P, P, P, P, P, I32, I32, I32, P],
    "nrhip_dec_bn_finalize": [P, I32, I64, P, P, F32, F32, P, P, P, P],
    "nrhip_dec_bn_act": [P, P, P, P, I64, P],
    "nrhip_dec_bn_bwd_workspace": [I64, C.POINTER(I64)],
    "nrhip_dec_bn_bwd": [P, P, P, P, P, P, P, P, P, P, I64, P],
    "nrhip_dec_grad_scale": [P, I64, P, P],
    "nrhip_dec_add_masked": [P, P, P, P, I64, P],
    "nrhip_dec_conv1x1_in_fwd": [P, P, P, P, I64, I32, P],
    "nrhip_dec_conv1x1_in_bwd_workspace": [I64, I32, C.POINTER(I64)],
    "nrhip_dec_conv1x1_in_bwd": [P, P, P, P, P, P, P, P, P, I64, I32, P],
    "nrhip_dec_upsample_pack": [P, P, P],
    "nrhip_dec_upsample_fwd": [P, P, P, P, I32, I32, I32, P],
    "nrhip_dec_upsample_bwd_workspace": [I32, I32, I32, C.POINTER(I64)],
    "nrhip_dec_upsample_bwd": [P, P, P, P, P, P, P, P, I32, I32, I32, P],
    "nrhip_dec_rgb_fwd": [P, P, P, P, I64, P],
    "nrhip_dec_rgb_bwd_workspace": [I64, C.POINTER(I64)],
    "nrhip_dec_rgb_bwd": [P, P, P, P, P, P, P, P, P, I64, P],
    "nrhip_rgb_decoder_sizes": [C.POINTER(RgbDecoder), C.POINTER(I64), C.POINTER(I64), C.POINTER(I64)],
    "nrhip_rgb_decoder_fwd": [C.POINTER(RgbDecoder), P, P, P, P, P],
    "nrhip_rgb_decoder_bwd": [C.POINTER(RgbDecoder), P, P, P, P, P, P, P, P],
    "nrhip_hashgrid_fwd": [C.POINTER(Grid), P, P, I64, P, P],
    "nrhip_hashgrid_bwd": [C.POINTER(Grid), P, P, I64, P, P],
    "nrhip_hashgrid_bwd_input": [C.POINTER(Grid), P, P, P, I64, P, P],
    "nrhip_hashgrid_multi_fwd": [C.POINTER(Grid), P, I32, P, P, I64, P, P],
    "nrhip_hashgrid_multi_bwd": [C.POINTER(Grid), I32, P, P, P, I64, P, P],
    "nrhip_hashgrid_multi_bwd_input": [C.POINTER(Grid), P, I32, P, P, P, I64, P, P],
    "nrhip_hashgrid_multi_bwd_binned_workspace": [C.POINTER(Grid), I32, I64, C.POINTER(I64)],
    "nrhip_hashgrid_multi_bwd_binned": [C.POINTER(Grid), I32, P, P, I32, P, P, I64, P, I32, P, I64, P],
    "nrhip_encode_fwd": [C.POINTER(Grid), P, F32, C.POINTER(Rays), P, P],
    "nrhip_encode_bwd": [C.POINTER(Grid), F32, C.POINTER(Rays), P, P, P],
    "nrhip_encode_bwd_rays": [C.POINTER(Grid), P, F32, C.POINTER(Rays), P, P, P, P],
    "nrhip_encode_bwd_binned_workspace": [C.POINTER(Grid), I64, C.POINTER(I64)],
    "nrhip_encode_bwd_binned": [C.POINTER(Grid), F32, C.POINTER(Rays), P, P, I32, P, I64, P],
    "nrhip_encode_bwd_binned_f16": [C.POINTER(Grid), F32, C.POINTER(Rays), P, P, P, I64, P],
    "nrhip_hashgrid_bwd_binned": [C.POINTER(Grid), P, P, I64, P, I32, P, I64, P],
    "nrhip_sh4_fwd": [P, I64, P, P],
    "nrhip_mlp_fwd": [C.POINTER(Mlp), P, I64, P, P, P],
    "nrhip_mlp_bwd_workspace": [C.POINTER(Mlp), I64, C.POINTER(I64)],
    "nrhip_mlp_bwd": [C.POINTER(Mlp), P, P, P, I64, P, C.POINTER(P), C.POINTER(P), P, I64, P],
    "nrhip_field_feature_bwd": [C.POINTER(Mlp), P, P, P, P, I64, P, C.POINTER(P), C.POINTER(P), P, I64, P],
    "nrhip_field_fwd": [C.POINTER(Field), C.POINTER(Rays), P, P, P, P],
    "nrhip_field_fwd_train": [C.POINTER(Field), C.POINTER(Rays), P, P, P, P, P, P, P, P],
    "nrhip_field_fwd_train_ovr": [C.POINTER(Field), C.POINTER(Rays), P, P, P, P, P, P, P, P, P, P, P],
    "nrhip_render_weight_from_alpha": [P, I64, I32, P, P, P],
    "nrhip_render_weight_from_alpha_bwd": [P, P, P, I64, I32, P, P],
    "nrhip_render_weight_from_density": [P, P, P, I64, I32, P, P, P, P],
    "nrhip_render_weight_from_density_bwd": [P, P, P, P, I64, I32, P, P],
    "nrhip_accumulate_along_rays": [P, P, I64, I32, I32, P, P],
    "nrhip_lidar_carving": [P, P, I32, P, P, P, P, F32, F32, I64, I32, P, P, P, P],
    "nrhip_embedding_lerp_fwd": [P, P, P, P, I64, I32, I32, P, P],
    "nrhip_embedding_lerp_bwd": [P, P, P, P, I64, I32, I32, P, P],
    "nrhip_accumulate_along_rays_bwd": [P, P, P, I64, I32, I32, P, P, P],
    "nrhip_composite_fwd": [P, P, P, P, I64, I32, I32, P, P, P, P],
    "nrhip_composite_bwd": [P, P, P, P, P, P, P, I64, I32, I32, P, P, P],
    "nrhip_render_fwd": [C.POINTER(Field), C.POINTER(Rays), P, P, P, P, P],
    "nrhip_render_fwd_ex": [C.POINTER(Field), C.POINTER(Rays), P, P, P, P, F32, P],
    "nrhip_ray_order": [P, P, I64, F32, F32, I32, P, P],
    "nrhip_ray_order_workspace": [I64, I32, C.POINTER(I64)],
    "nrhip_ray_order_large": [P, P, I64, F32, F32, I32, P, I64, P, P],
    "nrhip_camera_rays": [C.POINTER(CameraTable), P, P, I64, P, P, P, P, P, P],
    "nrhip_lidar_rays": [C.POINTER(LidarTable), P, P, I32, I64, P, P, P, P, P, P, P],
    "nrhip_patch_sample": [P, P, I64, I32, I32, I32, I32, I32, I32, P, P, I32, P, P, P, P],
    "nrhip_lidar_point_sample": [P, P, P, P, P, I32, I32, I32, I64, P, P, P],
    "nrhip_tuning_reload": [],
    "nrhip_adam_step": [P, P, P, P, I64, I64, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, P],
    "nrhip_adam_step_many": [C.POINTER(AdamTensor), I32, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double,
                             P],
    "nrhip_adam_step_many_workspace": [I32, C.POINTER(I64)],
    "nrhip_adam_step_many_dev": [C.POINTER(AdamTensorDev), I32, C.c_double, P, C.c_double, C.c_double, C.c_double, C.c_double,
                                 C.c_double, P, P, P, P],
    "nrhip_nonfinite_check_many": [C.POINTER(CheckTensor), I32, P, P],
    "nrhip_proposal_density_fwd": [C.POINTER(Proposal), C.POINTER(Rays), P, P, P],
    "nrhip_proposal_density_bwd": [C.POINTER(Proposal), C.POINTER(Rays), P, P, P, P, P],
    "nrhip_proposal_density_bwd_binned": [C.POINTER(Proposal), C.POINTER(Rays), P, P, P, P, P, I32, P, I64, P],
    "nrhip_weights_from_density": [P, P, I64, I32, P, P],
    "nrhip_weights_from_density_bwd": [P, P, P, I64, I32, P, P],
    "nrhip_power_sampler": [P, P, I64, I32, F32, F32, P, F32, P, P, P],
    "nrhip_power_sampler_ordered": [P, P, I64, I32, F32, F32, P, F32, P, P, P, P, F32, F32, I32, P, P],
    "nrhip_pdf_sample": [P, P, P, P, I64, I32, I32, F32, F32, F32, P, I32, P, P, P],
    "nrhip_actor_prepare": [C.POINTER(Actors), C.POINTER(Rays), P, P, P, P, P, P],
    "nrhip_actor_prepare_edited": [C.POINTER(Actors), C.POINTER(Rays), P, C.POINTER(ActorEdit), P, P, P, P, P],
    "nrhip_actor_encode": [C.POINTER(Actors), C.POINTER(Rays), P, P, P, I32, P, P, P, P, P],
    "nrhip_actor_hits": [C.POINTER(Actors), C.POINTER(Rays), P, P, P, P, P],
    "nrhip_actor_density": [C.POINTER(Actors), C.POINTER(Rays), P, P, P, P, I32, P, P, P, P],
    "nrhip_actor_pair_positions_fwd": [C.POINTER(Actors), C.POINTER(Rays), P, P, P, P, I64, P, P, P],
    "nrhip_actor_pair_positions_bwd": [C.POINTER(Actors), C.POINTER(Rays), P, P, P, P, I64, P, P, P, P, P],
    "nrhip_actor_pair_positions_bwd_rays": [C.POINTER(Actors), C.POINTER(Rays), P, P, P, P, I64, P, P, P, P, P, P, P],
    "nrhip_actor_pairs_count": [P, I64, P, P, P],
    "nrhip_actor_pairs_write": [P, I64, P, P, P, P, P],
    "nrhip_actor_density_splice_fwd": [P, I32, P, P, P, I64, P, P, P],
    "nrhip_actor_density_splice_bwd": [P, I32, P, P, P, P, P, P, I64, P, P, P, P],
    "nrhip_render_fwd_actors": [C.POINTER(Field), C.POINTER(Actors), C.POINTER(Rays), P, P, P, P, P, P, P, F32, P, P],
    "nrhip_occgrid_march": [C.POINTER(OccGrid), P, P, P, P, P, I64, F32, F32, F32, F32, I32, P, P, P, P, P, P],
    "nrhip_packed_visibility_from_alpha": [P, P, I64, F32, F32, P, P],
    "nrhip_proposal_sampler_fwd": [C.POINTER(SamplerCfg), C.POINTER(Proposal), P, P, P, P, P, I64, C.POINTER(P),
                                   C.POINTER(P), C.POINTER(P), P],
    "nrhip_proposal_sampler_fwd_actors": [C.POINTER(SamplerCfg), C.POINTER(Proposal), C.POINTER(Actors), P, P, P, P, P, P, P,
                                          P, I64, C.POINTER(P), C.POINTER(P), C.POINTER(P), P],
    "nrhip_interlevel_loss": [P, P, I32, P, P, I32, F32, I64, P, P, P],
    "nrhip_distortion_loss": [P, P, I32, I64, P, P, P],
    "nrhip_prop_weights_fwd": [P, I32, P, I64, I32, P, P, P],
    "nrhip_prop_weights_bwd": [P, I32, P, P, P, I64, I32, P, P],
    "nrhip_sdf_render_fwd": [P, P, F32, P, P, I32, I64, I32, I32, P, P, P, I32, P, P, P],
    "nrhip_sdf_render_bwd_workspace": [I64, C.POINTER(I64)],
    "nrhip_sdf_render_bwd": [P, P, F32, P, P, P, I32, P, I32, P, P, P, I64, I32, I32, P, P, P, P, P],
    "nrhip_appearance_fwd": [P, P, P, F32, I32, I32, I64, I32, I32, P, I32, P],
    "nrhip_appearance_bwd": [P, I32, P, P, F32, I32, I32, I64, I32, I32, P, P],
    "nrhip_mask_compact": [P, I64, P, I64, P, P, P],
    "nrhip_grad_rows_count": [P, I32, I64, I32, P, P, P],
    "nrhip_grad_rows_compact": [P, I32, I64, I32, P, C.POINTER(I32), C.POINTER(I64), I32, F32, P, P, P],
    "nrhip_grad_rows_apply": [P, I32, I64, I32, C.POINTER(I32), C.POINTER(I64), I32, P, P, I32, P],
    "nrhip_lidar_losses": [C.POINTER(P), I32, P, P, P, P, P, P, I64, F32, F32, F32, P, P, P, P],
    "nrhip_lidar_losses_workspace": [I64, C.POINTER(I64)],
    "nrhip_lidar_losses_bwd": [P, P, P, P, P, I32, I64, I64, C.POINTER(P), P, P, P],
}

_lib = None


class NeuradHipError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load the shared library (once) and bind every prototype; raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NeuradHipError(
            f"libneurad_hip.so not found at {LIB_PATH}: build it with `python __graft_entry__.py` "
            "(hipcc --offload-arch=gfx950).  There is no CPU/PyTorch fallback for the product path.")
    lib = C.CDLL(LIB_PATH)
    lib.nrhip_last_error.restype = C.c_char_p
    lib.nrhip_last_error.argtypes = []
    for name, argtypes in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.argtypes = argtypes
        fn.restype = C.c_int
    _lib = lib
    return lib


def call(name: str, *args) -> None:
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise NeuradHipError(f"{name} failed (code {rc}): {lib.nrhip_last_error().decode()}")
