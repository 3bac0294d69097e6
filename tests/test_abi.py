"""CPU: the C-ABI library loads and exports every symbol include/*.h declares."""
import ctypes
import glob
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = []
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        text = re.sub(r"/\*.*?\*/", "", open(h).read(), flags=re.S)
        names += re.findall(r"\b(dfx_[a-z0-9_]+)\s*\(", text)
    return sorted(set(names))


def test_header_declares_the_operator_entry_points():
    names = declared_symbols()
    for must in ("dfx_msda_forward_f32", "dfx_msda_forward_f64", "dfx_msda_backward_f32",
                 "dfx_msda_backward_f64", "dfx_msda_fused_forward_f32", "dfx_abi_version", "dfx_last_error"):
        assert must in names


def test_library_exports_every_declared_symbol():
    from dfx import _lib
    assert os.path.exists(_lib.library_path()), "libdfx.so not built (python __graft_entry__.py build)"
    lib = ctypes.CDLL(_lib.library_path())
    for name in declared_symbols():
        assert hasattr(lib, name), f"{name} declared in include/ but not exported"
    lib.dfx_abi_version.restype = ctypes.c_int
    assert lib.dfx_abi_version() >= 1


def test_python_binding_covers_the_header():
    from dfx import _lib
    bound = set(_lib.SIGNATURES) | {"dfx_abi_version", "dfx_last_error"}
    assert set(declared_symbols()) <= bound


def test_cpu_tensors_are_rejected_like_the_reference():
    """ms_deform_attn.h:38 - AT_ERROR("Not implemented on the CPU")."""
    import torch
    import MultiScaleDeformableAttention as MSDA
    v = torch.zeros(1, 4, 8, 32)
    s = torch.tensor([[2, 2]])
    l = torch.tensor([0])
    loc = torch.zeros(1, 3, 8, 1, 4, 2)
    aw = torch.zeros(1, 3, 8, 1, 4)
    with pytest.raises(RuntimeError, match="CPU"):
        MSDA.ms_deform_attn_forward(v, s, l, loc, aw, 64)
    with pytest.raises(RuntimeError, match="CPU"):
        MSDA.ms_deform_attn_backward(v, s, l, loc, aw, torch.zeros(1, 3, 256), 64)


def test_module_has_reference_surface():
    from models.ops.modules import MSDeformAttn
    m = MSDeformAttn(256, 1, 8, 4)
    keys = set(m.state_dict())
    assert keys == {f"{n}.{p}" for n in ("sampling_offsets", "attention_weights", "value_proj", "output_proj")
                    for p in ("weight", "bias")}
    assert m.im2col_step == 64
    assert m.sampling_offsets.weight.shape == (64, 256) and m.attention_weights.weight.shape == (32, 256)


def test_entry_points_reject_bad_arguments_before_any_launch():
    """Argument validation returns an error code without touching the GPU (no compute call is made here):
    activation codes outside 0..2, kernel sides beyond the 32-bit tap masks of the implicit GEMM."""
    from dfx import _lib
    lib = _lib.load()
    one = 16       # any non-null, 16-byte aligned address: the checks below fail before it would be used
    rc = lib.dfx_gemm_f32(one, 0, 4, 0, one, 4, 0, 0, 0, 0, 0, 0, 0, 0, 0, one, 4, 0, 4, 4, 4, 1, 3, 0, 0, 0, None)
    assert rc != 0 and b"activation" in lib.dfx_last_error()
    rc = lib.dfx_gemm_splitk_f32(one, 4, one, 4, 0, 0, 0, 0, 0, one, 4, 4, 4, 4, 7, 2, one, None)
    assert rc != 0 and b"activation" in lib.dfx_last_error()
    rc = lib.dfx_conv2d_igemm_f32(one, one, one, 0, one, 1, 4, 8, 40, 4, 8, 8, 144, 1, 33, 1, 0, 1, 0, 0, None)
    assert rc != 0 and b"kernel sides" in lib.dfx_last_error()
