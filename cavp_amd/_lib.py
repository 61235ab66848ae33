"""ctypes binding of libcavp_hip.so (include/cavp_hip.h).  There is NO fallback: if the library is missing or a
symbol does not resolve, importing the compute path raises."""
from __future__ import annotations

import ctypes as C
import os

# torch ships its own libamdhip64 (SONAME libamdhip64.so.7, found through its RPATH).  It must be in the process
# BEFORE libcavp_hip.so is dlopen'ed so that our NEEDED libamdhip64.so.7 binds to that same runtime instance;
# the other order loads /opt/rocm's copy as a second HIP runtime and every launch on a torch stream fails.
import torch  # noqa: F401  (ordering dependency, see above)

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libcavp_hip.so")

F32, BF16 = 0, 1
ACT_NONE, ACT_RELU, ACT_LEAKY, ACT_GELU = 0, 1, 2, 3
ABI_VERSION = 12
ERR_BAD_ARG, ERR_UNSUPPORTED, ERR_ALIGN, ERR_WORKSPACE, ERR_LAUNCH = -1, -2, -3, -4, -5   # cavp_status_t
WGRAD_GROUP_MAX = 16   # CAVP_WGRAD_GROUP_MAX


class CavpError(RuntimeError):
    pass


class ConvDesc(C.Structure):
    """struct cavp_conv_desc (include/cavp_hip.h)."""
    _fields_ = [(n, C.c_int32) for n in (
        "dtype", "N", "H", "W", "Cin", "ldx", "Cout", "ldy", "KH", "KW", "stride", "pad", "dil", "ldr", "act",
        "splitk", "tile", "up", "Ho", "Wo", "stride_w", "dw_oihw", "dw_overwrite", "res_rows", "aux_mode", "ld_aux")]


class WgradJob(C.Structure):
    """struct cavp_wgrad_job (include/cavp_hip.h)."""
    _fields_ = [("desc", ConvDesc), ("x", C.c_void_p), ("dy", C.c_void_p), ("dw", C.c_void_p), ("dbias", C.c_void_p)]


class BnBwdArgs(C.Structure):
    """struct cavp_bnbwd_args (include/cavp_hip.h)."""
    _fields_ = [("z", C.c_void_p), ("out", C.c_void_p), ("ld_z", C.c_int32), ("ld_out", C.c_int32), ("fwd_scale", C.c_void_p),
                ("fwd_shift", C.c_void_p), ("mean", C.c_void_p), ("rstd", C.c_void_p), ("act", C.c_int32), ("pad_", C.c_int32),
                ("partials", C.c_void_p)]


_vp, _i32, _i64, _f32, _sz = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_size_t

# name -> (restype, argtypes): every symbol include/cavp_hip.h declares
PROTOTYPES = {
    "cavp_abi_version": (_i32, []),
    "cavp_set_deterministic": (_i32, [_vp, _sz]),
    "cavp_zero_ranges_f32": (_i32, [_vp, _vp, _i32, _i64, _vp]),
    "cavp_zero_bytes": (_i32, [_vp, _sz, _vp]),
    "cavp_i64_add_table": (_i32, [_vp, _i32, _i64, _vp]),
    "cavp_get_deterministic": (_i32, []),
    "cavp_error_string": (C.c_char_p, [_i32]),
    "cavp_conv2d_workspace_bytes": (_sz, [C.POINTER(ConvDesc)]),
    "cavp_conv2d_nhwc": (_i32, [C.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp, _vp]),
    "cavp_conv2d_nhwc_aux": (_i32, [C.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp, _vp]),
    "cavp_set_tail_split": (_i32, [_i32]),
    "cavp_set_wgrad_big": (_i32, [_i32, _i32]),
    "cavp_attn1_supported": (_i32, [_i32, _i32]),
    "cavp_attn1_prepare": (_i32, [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _f32, _vp]),
    "cavp_attn1_fwd": (_i32, [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "cavp_attn1_bwd_workspace_bytes": (_sz, [_i32, _i32, _i32, _i32, _i32]),
    "cavp_attn1_bwd": (_i32, [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _i32, _i32, _i32, _i32, _i32, _vp]),
    "cavp_attn1_finish": (_i32, [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _f32, _vp]),
    "cavp_conv2d_tile_stats_layout": (_i32, [C.POINTER(ConvDesc), C.POINTER(_i32), C.POINTER(_i32)]),
    "cavp_bn_finalize_tiles": (_i32, [_vp, _i32, _i32, _i64, _vp, _vp, _f32, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp]),
    "cavp_bn_tiles_to_moments": (_i32, [_vp, _i32, _i32, _i64, _vp, _i32, _vp]),
    "cavp_conv3x3_smallcin_nchw": (_i32, [_i32, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "cavp_maxpool_nhwc": (_i32, [_i32, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "cavp_maxpool_affine_nhwc": (_i32, [_i32, _vp, _vp, _vp, _i32, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "cavp_global_avgpool_nhwc": (_i32, [_i32, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "cavp_bilinear_nhwc": (_i32, [_i32, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "cavp_bilinear_nhwc_to_nchw": (_i32, [_i32, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "cavp_layernorm": (_i32, [_i32, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _f32, _vp]),
    "cavp_layernorm_residual": (_i32, [_i32, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _f32, _vp]),
    "cavp_attn_gate": (_i32, [_i32, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _f32, _i32, _vp]),
    "cavp_bn_fold": (_i32, [_vp, _vp, _vp, _vp, _f32, _vp, _vp, _i32, _vp]),
    "cavp_pack_weight_ohwi": (_i32, [_i32, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "cavp_cast": (_i32, [_i32, _vp, _i32, _vp, _i64, _vp]),
    # ---- training side ----
    "cavp_conv2d_wgrad_workspace_bytes": (_sz, [C.POINTER(ConvDesc)]),
    "cavp_conv2d_wgrad_nhwc": (_i32, [C.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "cavp_conv2d_wgrad_group_workspace_bytes": (_sz, [_vp, _i32]),
    "cavp_conv2d_wgrad_group": (_i32, [_vp, _i32, _vp, _sz, _vp]),
    "cavp_pack_weight_dgrad": (_i32, [_i32, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "cavp_pack_weights_multi": (_i32, [_i32, _vp, _i32, _vp]),
    "cavp_optimizer_blocks": (_i32, [C.c_int64]),
    "cavp_optimizer_step": (_i32, [_vp, _i32, _i32, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                   C.c_int64, _vp]),
    "cavp_mel_frontend": (_i32, [_vp, _i32, _i32, _vp, _vp, _vp, _i32, _i32, _i32, _i32, C.c_float, C.c_float, C.c_float, _vp]),
    "cavp_unpack_weight_grad": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "cavp_conv3x3_smallcin_wgrad": (_i32, [_i32, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, C.c_size_t, _vp]),
    "cavp_conv3x3_smallcin_wgrad_workspace_bytes": (C.c_size_t, [_i32, _i32, _i32, _i32, _i32, _i32, _i32]),
    "cavp_colstats": (_i32, [_i32, _vp, _vp, _i64, _i32, _i32, _vp, _vp, _vp]),
    "cavp_scale_f32": (_i32, [_vp, _f32, _vp, _i32, _vp]),
    "cavp_bn_finalize": (_i32, [_vp, _vp, _vp, _i64, _vp, _vp, _f32, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp]),
    "cavp_conv2d_bnbwd_layout": (_i32, [_vp, _vp, _vp]),
    "cavp_conv2d_nhwc_bnbwd": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "cavp_bn_bwd_sum_tiles": (_i32, [_vp, _i32, _i32, _vp, _vp, _vp]),
    "cavp_scale_shift_act": (_i32, [_i32, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _vp]),
    "cavp_bn_act_bwd_reduce": (_i32, [_i32, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp,
                                      _vp]),
    "cavp_bn_act_bwd_apply": (_i32, [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32,
                                     _vp, _i32, _vp, _i32, _vp, _vp, _vp]),
    "cavp_bn_act_bwd_apply_acc": (_i32, [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32,
                                         _vp, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _vp]),
    "cavp_act_bwd": (_i32, [_i32, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _vp]),
    "cavp_add": (_i32, [_i32, _vp, _vp, _vp, _i64, _vp]),
    "cavp_colsum": (_i32, [_i32, _vp, _i64, _i32, _i32, _vp, _vp]),
    "cavp_colsum_groups": (_i32, [_i32, _vp, _i32, _i32, _i32, _i32, _vp, _vp]),
    "cavp_col_tile_stats": (_i32, [_i32, _vp, _i64, _i32, _i32, _vp, _vp]),
    "cavp_layernorm_bwd": (_i32, [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _vp]),
    "cavp_layernorm_bwd_add": (_i32, [_i32, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _i32, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _f32,
                                      _vp]),
    "cavp_attn_gate_bwd": (_i32, [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _f32, _i32, _vp]),
    "cavp_maxpool_bwd_nhwc": (_i32, [_i32, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "cavp_bilinear_bwd_nhwc": (_i32, [_i32, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "cavp_bilinear_bwd_nchw_to_nhwc": (_i32, [_i32, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "cavp_bcast_add_nhwc": (_i32, [_i32, _vp, _vp, _f32, _i32, _i32, _i32, _i32, _vp]),
    "cavp_ce_loss_nchw": (_i32, [_vp, _vp, _i32, _i32, _i32, _i64, _i32, _f32, _vp, _vp, _vp, _vp]),
    "cavp_upsample_ce_head": (_i32, [_i32, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _vp, _vp,
                              _vp, _vp, _vp]),
    # ---- contrastive loss ----
    "cavp_label_nearest": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "cavp_gather_l2norm": (_i32, [_vp, _i64, _i64, _i64, _vp, _vp, _i32, _i32, _f32, _vp, _vp, _vp]),
    "cavp_infonce_rows": (_i32, [_vp, _vp, _i32, _i32, _f32, _vp, _vp, _vp, _f32, _vp]),
    "cavp_symm_add": (_i32, [_vp, _vp, _i32, _f32, _vp]),
    "cavp_symm_add_scaled": (_i32, [_vp, _vp, _i32, _f32, _vp, _vp]),
    "cavp_l2norm_bwd_scatter": (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _i64, _i64, _i64, _vp]),
    # ---- PVTv2 ----
    "cavp_sra_attention": (_i32, [_i32, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _vp]),
    "cavp_dwconv3x3_nhwc": (_i32, [_i32, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "cavp_dwconv3x3_nhwc_aux": (_i32, [_i32, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "cavp_dwconv3x3_bwd_data_nhwc": (_i32, [_i32, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "cavp_pack_dwconv_weight": (_i32, [_vp, _vp, _i32, _vp]),
    "cavp_conv_smallcin_kxk_nchw": (_i32, [_i32, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "cavp_sra_attention_bwd_workspace_bytes": (_sz, [_i32, _i32, _i32]),
    "cavp_sra_attention_bwd": (_i32, [_i32, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _vp, _sz, _vp]),
    "cavp_sra_attention_bwd_to": (_i32, [_i32, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _vp, _sz, _vp]),
    "cavp_dwconv3x3_wgrad": (_i32, [_i32, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "cavp_dwconv3x3_bwd": (_i32, [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "cavp_smallcin_kxk_im2col": (_i32, [_i32, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "cavp_space_to_depth": (_i32, [_i32, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "cavp_row_scale_add": (_i32, [_i32, _vp, _vp, _vp, _vp, _i32, _i64, _vp]),
}

_lib = None


def load() -> C.CDLL:
    """dlopen the library and bind every prototype.  Raises CavpError (never falls back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CavpError(
            f"{LIB_PATH} not found: the CAVP MI355X path has no CPU/PyTorch fallback. Build it with "
            f"`python -m cavp_amd.build` (needs hipcc, cross-compiles gfx950 without a GPU).")
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:  # e.g. libamdhip64 missing
        raise CavpError(f"cannot dlopen {LIB_PATH}: {e}") from e
    for name, (res, args) in PROTOTYPES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise CavpError(f"{LIB_PATH} does not export {name}; rebuild with `python -m cavp_amd.build --force`") from e
        fn.restype, fn.argtypes = res, args
    v = lib.cavp_abi_version()
    if v != ABI_VERSION:
        raise CavpError(f"libcavp_hip.so ABI {v} != expected {ABI_VERSION}; rebuild")
    _lib = lib
    return lib


def check(status: int, what: str = "") -> None:
    if status != 0:
        msg = load().cavp_error_string(status).decode()
        raise CavpError(f"{what}: libcavp_hip status {status} ({msg})")


_det_scratch = None


def set_deterministic(on: bool, device=None, scratch_bytes: int = 8 << 20) -> None:
    """Opt-in bit-reproducible training (the reference runs with cudnn.deterministic = True, main_vpo_mono.py:39-41): the
    reductions that otherwise end in f32 atomics add their per-workgroup partials in a fixed order (cavp_set_deterministic,
    include/cavp_hip.h).  Call it BEFORE capturing a training step into a hipGraph: the scratch buffer's address is baked into
    the captured launches.  Costs ~0.3 ms per C1' step (a few dozen extra small launches)."""
    global _det_scratch
    import torch
    lib = load()
    if on:
        _det_scratch = torch.empty(scratch_bytes, dtype=torch.uint8, device=device if device is not None else torch.device("cuda", torch.cuda.current_device()))
        check(lib.cavp_set_deterministic(C.c_void_p(_det_scratch.data_ptr()), C.c_size_t(scratch_bytes)), "cavp_set_deterministic")
    else:
        check(lib.cavp_set_deterministic(None, C.c_size_t(0)), "cavp_set_deterministic")
        _det_scratch = None   # (graphs captured while it was on must not be replayed any more)


def is_deterministic() -> bool:
    return bool(load().cavp_get_deterministic())
