"""ctypes binding of libtd_engine.so (C-ABI declared in include/td_engine.h).

There is NO CPU fallback: if the HIP library is missing or no GPU is visible, calls raise.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libtd_engine.so")


class UnetConfig(C.Structure):
    _fields_ = [("image_size", C.c_int32), ("in_channels", C.c_int32), ("out_channels", C.c_int32),
                ("model_channels", C.c_int32), ("n_levels", C.c_int32), ("channel_mults", C.c_int32 * 8),
                ("layers_per_block", C.c_int32 * 8), ("n_attn_resolutions", C.c_int32),
                ("attn_resolutions", C.c_int32 * 8), ("midblock_attention", C.c_int32), ("concat_balance", C.c_float),
                ("noise_emb_dims", C.c_int32), ("emb_channels", C.c_int32), ("n_cond", C.c_int32),
                ("cond_type", C.c_int32 * 8), ("cond_dims", C.c_int32 * 8), ("cond_weights", C.c_float * 8)]


class TdError(RuntimeError):
    pass


_P = C.c_void_p
_SIGS = {
    "td_last_error": (C.c_char_p, []),
    "td_version": (C.c_int, []),
    "td_build_id": (C.c_char_p, []),
    "td_engine_create": (C.c_int, [C.c_int, C.POINTER(_P)]),
    "td_engine_destroy": (None, [_P]),
    "td_engine_synchronize": (C.c_int, [_P]),
    "td_engine_stream": (_P, [_P]),
    "td_engine_set_stream": (C.c_int, [_P, _P]),
    "td_engine_set_option": (C.c_int, [_P, C.c_char_p, C.c_int64]),
    "td_engine_profile_read": (C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_int64), C.c_int]),
    "td_engine_profile_read_glds": (C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64), C.c_int]),
    "td_engine_profile_dump": (C.c_int, [_P, C.c_char_p, C.c_int64]),
    "td_unet_create": (C.c_int, [_P, C.POINTER(UnetConfig), C.c_int, C.POINTER(_P)]),
    "td_unet_destroy": (None, [_P]),
    "td_unet_num_params": (C.c_int, [_P]),
    "td_unet_param_info": (C.c_int, [_P, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int32), C.POINTER(C.c_int64 * 4)]),
    "td_unet_set_param": (C.c_int, [_P, C.c_char_p, _P, C.c_int64]),
    "td_unet_set_prefolded": (C.c_int, [_P, C.c_int]),
    "td_unet_finalize": (C.c_int, [_P]),
    "td_unet_cond_row_len": (C.c_int, [_P]),
    "td_unet_forward": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P]),
    "td_unet_read_activation": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_char_p, _P, C.c_int64, C.POINTER(C.c_int32 * 4)]),
    "td_tile_seed": (C.c_uint64, [C.c_uint64, C.c_int64, C.c_int64]),
    "td_standard_normal": (C.c_int, [_P, C.c_uint64, C.c_int64, _P]),
    "td_noise_patches": (C.c_int, [_P, C.c_uint64, C.c_int, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, _P]),
    "td_sample_edm": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_float, _P, _P]),
    "td_sample_edm_guided": (C.c_int, [_P, _P, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_float, _P, _P]),
    "td_sample_consistency": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, _P, _P, _P, _P]),
    "td_sample_edm_img": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_float, _P, _P, C.c_int, _P]),
    "td_sample_consistency_img": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, _P, _P, _P, _P, C.c_int, _P]),
    "td_blend_windows": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_int, _P, C.c_int, _P, _P, _P, C.c_int]),
    "td_gather_regions": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_int, _P, _P]),
    "td_blend_normalize": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_float, _P]),
    "td_linear_weight_window": (C.c_int, [_P, C.c_int, _P]),
    "td_attention": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, _P]),
    "td_perlin_map": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_float, C.c_float, _P, _P, C.c_int, _P]),
    "td_resample2d": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, C.c_int, _P, _P, C.c_int, _P]),
    "td_residual_plus": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_float, C.c_float, _P]),
    "td_elev_finish": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, _P]),
    "td_ddim_cfg_step": (C.c_int, [_P, _P, _P, _P, C.c_int64, C.c_float, C.c_float, C.c_float, _P]),
    "td_climate_finish": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, _P]),
}
EXPORTS = tuple(_SIGS)

_lib = None


def lib():
    """Loads the HIP engine; raises if it has not been built (python __graft_entry__.py build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise TdError(f"{LIB_PATH} is missing: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()'). "
                          "There is no CPU fallback.")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(rc):
    if rc != 0:
        raise TdError(f"td_engine error {rc}: {lib().td_last_error().decode()}")
