"""ctypes binding of ``libsihl_hip.so`` (the C-ABI declared in ``include/sihl_hip.h``).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C sihl_amd/csrc``.  There is
no CPU fallback: ``lib()`` raises if the shared object is missing, and every op raises if it is
handed a tensor that is not on a HIP device.
"""
import ctypes
import os
from ctypes import c_float, c_int, c_long, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# SIHL_HIP_LIB: developer override (same-box A/B of two builds); the product always loads the in-tree library
LIB_PATH = os.environ.get("SIHL_HIP_LIB") or os.path.join(_HERE, "libsihl_hip.so")

F32, BF16 = 0, 1
ACT = {None: 0, "none": 0, "relu": 1, "silu": 2, "sigmoid": 3}

P, I, L, F = c_void_p, c_int, c_long, c_float

# name -> (restype, argtypes); order is exactly the C prototype in include/sihl_hip.h
SIGNATURES = {
    "sihl_conv2d_stat_rows": (I, [L]),
    "sihl_conv2d_force_register_staging": (I, [I]),
    "sihl_conv2d_tile_override": (I, [I]),
    "sihl_conv2d_nbuf_override": (I, [I]),
    "sihl_conv2d_debug": (I, [I]),
    "sihl_conv2d_strided_classes_enable": (I, [I]),
    "sihl_conv2d_fwd": (I, [P, P, P, P, I, I, I, I, I, I, I, I, I, I, I, I, P, P, P, P, I, P, L, L, P]),
    "sihl_od_loss_ws_bytes": (L, [L, I]),
    "sihl_od_loss": (I, [P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, L, I, I, P, P, P, P, P, I, P, L, P]),
    "sihl_mlp_fwd_supported": (I, [L, I, I, I, I, I, I]),
    "sihl_mlp_fwd": (I, [P, L, L, I, I, I, P, P, P, P, F, I, I, P, I, I, P]),
    "sihl_mlp_rows_supported": (I, [L, I, I, I, I, I, I]),
    "sihl_mlp_rows_fwd_multi": (I, [P, I, I, I, P]),
    "sihl_mlp_rows_config": (I, [I]),
    "sihl_mlp_rows_debug": (I, [I]),
    "sihl_mlp_permute_k": (I, [P, P, L, I, P]),
    "sihl_mlp_rows_fwd": (I, [P, L, L, I, I, I, P, P, P, P, F, I, I, P, I, I, P]),
    "sihl_mlp_stages": (I, [I]),
    "sihl_mlp_debug": (I, [I]),
    "sihl_mlp_stamps": (I, [P]),
    "sihl_conv2d_dgrad": (I, [P, P, P, I, I, I, I, I, I, I, I, I, I, I, P]),
    "sihl_conv2d_ws_bytes": (L, [I, I, I, I, I, I, I, I, I, I]),
    "sihl_conv2d_fwd_ws": (I, [P, P, P, P, I, I, I, I, I, I, I, I, I, I, I, I, P, P, P, P, I, P, L, L, P, L, P]),
    "sihl_conv2d_dgrad_ws": (I, [P, P, P, P, I, I, I, I, I, I, I, I, I, I, I, P, L, P]),
    "sihl_conv2d_dgrad_add": (I, [P, P, P, P, I, I, I, I, I, I, I, I, I, I, I, I, P, L, P]),
    "sihl_conv2d_splitk_enable": (I, [I]),
    "sihl_conv2d_small_enable": (I, [I]),
    "sihl_conv2d_small_mode": (I, []),
    "sihl_conv2d_halo_enable": (I, [I]),
    "sihl_conv2d_rules_off": (I, [I]),
    "sihl_conv2d_krot": (I, [I]),
    "sihl_conv2d_wgrad_force_register_staging": (I, [I]),
    "sihl_conv2d_wgrad_ws_bytes": (L, [I, I, I, I, I, I, I, I, I, I, I, I]),
    "sihl_conv2d_wgrad": (I, [P, P, P, I, I, I, I, I, I, I, I, I, I, I, I, I, P, L, P]),
    "sihl_weight_flip_transpose": (I, [P, P, I, I, I, I, I, I, I, P]),
    "sihl_weight_prepare": (I, [P, I, L, L, P]),
    "sihl_bn_finalize": (I, [P, I, I, L, P, P, F, F, P, P, P, P, P, P, P]),
    "sihl_bn_eval_affine": (I, [P, P, P, P, F, I, P, P, P]),
    "sihl_affine_act": (I, [P, P, L, I, P, P, I, I, P]),
    "sihl_affine_add_act": (I, [P, P, P, P, L, I, P, P, I, I, P]),
    "sihl_affine_act_bwd": (I, [P, P, P, L, I, P, P, I, I, P]),
    "sihl_bn_stats_rows": (I, [L, I, I]),
    "sihl_bn_stats": (I, [P, L, I, P, I, I, P]),
    "sihl_stem_xp_bytes": (L, [I, I, I]),
    "sihl_stem_stats_rows": (I, [I, I]),
    "sihl_stem_conv_fwd": (I, [P, I, L, L, L, L, P, L, L, L, L, P, P, P, P, I, I, I, P]),
    "sihl_stem_wgrad_parts": (I, [I, I]),
    "sihl_stem_conv_wgrad": (I, [P, P, P, L, L, L, L, P, I, I, I, P]),
    "sihl_norm_act_bwd_ws_bytes": (L, [L, I, I]),
    "sihl_norm_act_bwd": (I, [P, P, P, L, I, P, P, P, P, P, P, I, I, I, I, P, L, P]),
    "sihl_norm_add_relu_bwd": (I, [P, P, P, P, P, P, L, I, P, P, P, P, P, P, I, I, P, L, P]),
    "sihl_fuse_up2": (I, [P, P, P, P, I, I, I, I, I, P]),
    "sihl_fuse_up2_bwd": (I, [P, P, P, P, P, P, P, P, I, I, I, I, I, P]),
    "sihl_nearest_up2_add": (I, [P, P, P, I, I, I, I, I, P]),
    "sihl_nearest_up2_add_bwd": (I, [P, P, I, I, I, I, I, P]),
    "sihl_resize_bilinear": (I, [P, P, P, I, I, I, I, I, I, I, P]),
    "sihl_resize_bilinear_bwd": (I, [P, P, I, I, I, I, I, I, I, P]),
    "sihl_add_act": (I, [P, P, P, L, I, I, P]),
    "sihl_maxpool3x3s2_fwd": (I, [P, P, P, I, I, I, I, I, P]),
    "sihl_maxpool3x3s2_bwd": (I, [P, P, P, I, I, I, I, I, P]),
    "sihl_fuse_sum": (I, [P, P, P, P, P, L, I, I, P]),
    "sihl_fuse_sum_bwd": (I, [P, P, P, P, P, P, P, P, P, P, L, I, I, P]),
    "sihl_blur_fuse": (I, [P, P, P, P, P, P, P, I, I, I, I, I, P]),
    "sihl_pyr_conv_supported": (I, [I, I, I, I, I]),
    "sihl_pyr_conv_stat_rows": (I, [I, I]),
    "sihl_pyr_conv_fwd": (I, [P, P, P, P, I, I, I, I, I, P, P, P, P, I, P, L, I, P, P, P, P, P, P, P, I, P, P, P, P, P]),
    "sihl_grad_clip": (I, [P, I, P, P, P, F, P, L, I, P]),
    "sihl_blur_fuse_bwd": (I, [P, P, P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, P]),
    "sihl_layernorm_act": (I, [P, P, L, I, P, P, F, I, P, P, I, P]),
    "sihl_layernorm_bwd_waves": (I, [L]),
    "sihl_layernorm_act_bwd_ws_bytes": (L, [L, I]),
    "sihl_layernorm_act_bwd": (I, [P, P, P, L, I, P, P, P, P, I, P, P, I, P, L, P]),
    "sihl_colsum_ws_bytes": (L, [L, I]),
    "sihl_colsum": (I, [P, L, I, P, I, P, L, P]),
    "sihl_topk_select_enable": (I, [I]),
    "sihl_topk_rows": (I, [P, I, I, I, I, P, P, I, P]),
    "sihl_gather_rows": (I, [P, P, P, I, I, I, I, I, P]),
    "sihl_od_decode": (I, [P, P, P, L, P, L, P, I, I, I, I, I, I, P, P, P, P, I, P]),
    "sihl_od_anchors": (I, [P, I, P, P, P]),
    "sihl_iseg_mask_decode": (I, [P, P, L, P, P, I, I, I, I, I, I, I, P, I, P]),
    "sihl_uafm_fwd": (I, [P, P, P, P, P, P, P, P, I, I, I, I, I, P]),
    "sihl_uafm_bwd_ws_bytes": (L, [I, I, I]),
    "sihl_uafm_bwd": (I, [P, P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, P, L, P]),
    "sihl_softmax_max_resize": (I, [P, P, P, I, I, I, I, I, I, I, P]),
    "sihl_ce_resize": (I, [P, P, L, P, P, P, I, I, I, I, I, I, I, P]),
    "sihl_profile_enable": (I, [I]),
    "sihl_profile_collect": (I, [I, I, P, P, P, P]),
    "sihl_profile_records": (L, [I, I, P, L]),
}

_lib = None


def lib() -> ctypes.CDLL:
    """Load the HIP library once; fail loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the sihl_amd hot path has no CPU fallback. "
                "Build it with `python -c 'import __graft_entry__ as g; g.build()'` or `make -C sihl_amd/csrc`.")
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the symbol is not exported
            fn.restype, fn.argtypes = res, args
        _lib = handle
    return _lib


class SihlHipError(RuntimeError):
    pass


def check(rc: int, what: str) -> None:
    if rc == 0:
        return
    if rc == -1:
        raise SihlHipError(f"{what}: invalid argument / unsupported shape (SIHL_EARG)")
    if rc == -2:
        raise SihlHipError(f"{what}: workspace too small (SIHL_EWS)")
    raise SihlHipError(f"{what}: hipError_t {rc}")
