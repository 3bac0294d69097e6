"""ctypes loader of libdfx.so (the C ABI declared in include/dfx_msda.h)."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# DFX_LIBRARY selects another build of the library (diagnostic / ablation builds of tools/ab.sh); it must exist - there is no
# fall back to the shipped one
_SO = os.environ.get("DFX_LIBRARY") or os.path.join(_HERE, "libdfx.so")
_lib = None

_i, _l, _p = ctypes.c_int, ctypes.c_long, ctypes.c_void_p
_DIMS = [_i] * 7  # N, S, M, D, L, Lq, P

# name -> argtypes; every function returns int (0 = ok, <0 = DFX_E*)
SIGNATURES = {
    "dfx_msda_forward_f32": [_p] * 5 + _DIMS + [_p, _p],
    "dfx_msda_forward_f64": [_p] * 5 + _DIMS + [_p, _p],
    "dfx_msda_backward_f32": [_p] * 6 + _DIMS + [_p, _p, _p, _p],
    "dfx_msda_backward_f64": [_p] * 6 + _DIMS + [_p, _p, _p, _p],
    "dfx_msda_fused_forward_f32": [_p, _p, _p, _p, _i, _i, _p, _l, _p, _l] + _DIMS + [_p, _p],
    "dfx_profile_enable": [_i],
    "dfx_tuning_reload": [],
    "dfx_profile_drain": [_p, _p, _p, _p, _i],
    "dfx_msda_fused_level_fits": [_i, _i],
    "dfx_msda_fused_level_forward_f32": [_p, _p, _i, _p, _p, _p, _i, _i, _i, _i, _p, _p],
    # include/dfx_mha.h: (ptr, batch stride, row stride) x (q, k, v, out), B, heads, Lq, Lk, scale, stream
    "dfx_mha_f32": [_p, _l, _l, _p, _l, _l, _p, _l, _l, _p, _l, _l, _i, _i, _i, _i, ctypes.c_float, _p],
    # include/dfx_roi.h: input, rois, N, C, H, W, K, ph, pw, scale, sampling_ratio, aligned, out, stream
    "dfx_roi_align_nchw_f32": [_p, _p] + [_i] * 7 + [ctypes.c_float, _i, _i, _p, _p],
    "dfx_roi_align_nhwc_f32": [_p, _p] + [_i] * 7 + [ctypes.c_float, _i, _i, _p, _p],
    "dfx_dynamic_conv_f32": [_p, _p, _l, _p, _p, _p, _p, _p, _i, _i, _i, _i, ctypes.c_float, _p],
    # include/dfx_preprocess.h
    "dfx_preprocess_u8_f32": [_p, _i, _i, _i, _p, _p, _i, _p, _p, _i, _i, _i, _p, _p, _p, _l, _i, _i, _p, _p],
    # include/dfx_gemm.h
    "dfx_gemm_f32": [_p, _p, _l, _l, _p, _l, _l, _i, _p, _i, _p, _l, _l, _p, _l, _p, _l, _l, _i, _i, _i, _i, _i, _i, _l, _l, _p],
    "dfx_gemm_splitk_f32": [_p, _l, _p, _l, _i, _p, _i, _p, _l, _p, _l, _i, _i, _i, _i, _i, _p, _p],
    "dfx_linear_ln_f32": [_p, _p, _l, _l, _p, _l, _p, _p, _l, _p, _p, ctypes.c_float, _p, _l, _i, _i, _i, _i, _p],
    "dfx_conv1x1_pair_f32": [_p, _p, _l, _i, _p, _l, _i, _p, _p, _l, _i, _i, _i, _i, _p],
    # include/dfx_fused.h: x, bias, residual, out, N, C, HW, relu, stream
    "dfx_bias_act_nchw_f32": [_p, _p, _p, _p, _i, _i, _l, _i, _p],
    "dfx_bias_relu_maxpool_f32": [_p, _p, _p, _i, _i, _i, _i, _p],
    "dfx_add_layernorm_f32": [_p, _p, _p, _p, _p, _l, _i, ctypes.c_float, _p],
    "dfx_box_refine_f32": [_p, _p, _i, _p, _l, ctypes.c_float, _p],
    "dfx_group_norm_f32": [_p, _p, _p, _p, _p, _i, _i, _l, _i, ctypes.c_float, _i, _p],
    # include/dfx_conv.h
    "dfx_conv2d_igemm_f32": [_p, _p, _p, _p, _p] + [_i] * 14 + [_l, _p],
    "dfx_conv2d_tile_fits": [_i] * 6,
    "dfx_conv2d_tile_f32": [_p, _p, _p, _p] + [_i] * 13 + [_l, _p],
    "dfx_conv3x3_wino_f32": [_p, _p, _p, _p] + [_i] * 7 + [_p],
    "dfx_wino_weights_f32": [_p, _p, _p, _i, _i, _p],
}


class LevelLayout(ctypes.Structure):
    """dfx_msda_level_layout (include/dfx_msda.h): operand strides of the level-in-LDS kernel, in floats."""
    _fields_ = [(n, ctypes.c_long) for n in (
        "value_frame", "value_token", "value_head", "value_oct", "value_chunk",
        "off_row", "off_head", "logit_row", "logit_head",
        "out_row", "out_head", "out_oct", "out_chunk")]


def library_path():
    return _SO


def load():
    """Return the loaded library; raise loudly if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            raise RuntimeError(
                f"{_SO} is missing: the HIP extension has not been built "
                "(run `python __graft_entry__.py build` or `make -C <pkg>/csrc`). "
                "There is no CPU or PyTorch fallback for this operator.")
        lib = ctypes.CDLL(_SO)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.argtypes = argtypes
            fn.restype = ctypes.c_int
        lib.dfx_abi_version.restype = ctypes.c_int
        lib.dfx_last_error.restype = ctypes.c_char_p
        _lib = lib
    return _lib


def abi_version():
    return load().dfx_abi_version()


def check(rc, what):
    if rc != 0:
        msg = load().dfx_last_error().decode() or f"error code {rc}"
        raise RuntimeError(f"{what}: {msg}")
