"""ctypes front end of the CPU oracle -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  The product package never does: its operators
fail loudly when the HIP library is missing instead of falling back to this.

The arithmetic lives in ``msda_oracle.c`` (each function cites the reference
file:line it follows).  This file only marshals torch/numpy CPU buffers and
reproduces the dimension derivation of the reference host code
(/root/reference/models/ops/src/cuda/ms_deform_attn_cuda.cu:40-48): L comes
from ``spatial_shapes.size(0)``, Lq from ``sampling_loc.size(1)``, P from
``sampling_loc.size(4)``; buffers are then read as flat arrays.
"""
import ctypes
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("DFX_ORACLE_LIBRARY") or os.path.join(_HERE, "libdfx_oracle.so")      # (override: a sanitizer build, tools/oracle_sanitize.sh)
_lib = None


def build(force=False):
    """Compile the C restatement with gcc (seconds)."""
    src = os.path.join(_HERE, "msda_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        for suf in ("f32", "f64"):
            getattr(_lib, f"dfx_oracle_msda_forward_{suf}").restype = ctypes.c_int
            getattr(_lib, f"dfx_oracle_msda_backward_{suf}").restype = ctypes.c_int
        _lib.dfx_oracle_roi_align_f32.restype = ctypes.c_int
        _lib.dfx_oracle_set_threads.restype = ctypes.c_int
    return _lib


def set_threads(n):
    return lib().dfx_oracle_set_threads(ctypes.c_int(int(n)))


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def _prep(value, shapes, lsi, loc, aw):
    assert value.device.type == "cpu"
    value = value.contiguous()
    loc = loc.contiguous().to(value.dtype)
    aw = aw.contiguous().to(value.dtype)
    shapes = shapes.contiguous().to(torch.int64)
    lsi = lsi.contiguous().to(torch.int64)
    N, S, M, D = value.shape
    L = shapes.shape[0]
    Lq = loc.shape[1]
    P = loc.shape[4]
    # the flat reads must stay inside the buffers the caller handed over
    assert loc.numel() >= N * Lq * M * L * P * 2, "sampling_loc smaller than the flat read"
    assert aw.numel() >= N * Lq * M * L * P, "attn_weight smaller than the flat read"
    suf = {torch.float32: "f32", torch.float64: "f64"}[value.dtype]
    return value, shapes, lsi, loc, aw, (N, S, M, D, L, Lq, P), suf


def msda_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight):
    """Oracle of MSDA.ms_deform_attn_forward; returns [N, Lq, M*D]."""
    value, shapes, lsi, loc, aw, dims, suf = _prep(value, spatial_shapes, level_start_index,
                                                   sampling_loc, attn_weight)
    N, S, M, D, L, Lq, P = dims
    out = torch.zeros(N, Lq, M * D, dtype=value.dtype)
    rc = getattr(lib(), f"dfx_oracle_msda_forward_{suf}")(
        _p(value), _p(shapes), _p(lsi), _p(loc), _p(aw),
        *[ctypes.c_int(v) for v in dims], _p(out))
    if rc != 0:
        raise RuntimeError(f"oracle forward failed rc={rc}")
    return out


def msda_backward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output):
    """Oracle of MSDA.ms_deform_attn_backward; returns (grad_value, grad_loc, grad_aw)."""
    value, shapes, lsi, loc, aw, dims, suf = _prep(value, spatial_shapes, level_start_index,
                                                   sampling_loc, attn_weight)
    go = grad_output.contiguous().to(value.dtype)
    gv = torch.zeros_like(value)
    gl = torch.zeros_like(loc)
    ga = torch.zeros_like(aw)
    rc = getattr(lib(), f"dfx_oracle_msda_backward_{suf}")(
        _p(value), _p(shapes), _p(lsi), _p(loc), _p(aw), _p(go),
        *[ctypes.c_int(v) for v in dims], _p(gv), _p(gl), _p(ga))
    if rc != 0:
        raise RuntimeError(f"oracle backward failed rc={rc}")
    return gv, gl, ga


class OracleMSDAFunction(torch.autograd.Function):
    """autograd wrapper with the argument list of the reference's
    MSDeformAttnFunction (models/ops/functions/ms_deform_attn_func.py:21-38);
    tests patch it over the product operator to run host logic on CPU."""

    @staticmethod
    def forward(ctx, value, shapes, lsi, loc, aw, im2col_step):
        ctx.save_for_backward(value, shapes, lsi, loc, aw)
        return msda_forward(value, shapes, lsi, loc, aw)

    @staticmethod
    def backward(ctx, grad_output):
        value, shapes, lsi, loc, aw = ctx.saved_tensors
        gv, gl, ga = msda_backward(value, shapes, lsi, loc, aw, grad_output)
        return gv, None, None, gl, ga, None


def roi_align(inp, rois, output_size=7, spatial_scale=1.0, sampling_ratio=2, aligned=True):
    """Oracle of the RoIAlign the TransVOD++ temporal stage uses (PARITY UNPINNED,
    see msda_oracle.c).  inp [N,C,H,W] fp32, rois [K,5] -> [K,C,ph,pw]."""
    inp = inp.contiguous().float()
    rois = rois.contiguous().float()
    N, C, H, W = inp.shape
    K = rois.shape[0]
    out = torch.zeros(K, C, output_size, output_size, dtype=torch.float32)
    rc = lib().dfx_oracle_roi_align_f32(
        _p(inp), _p(rois), *[ctypes.c_int(v) for v in (N, C, H, W, K, output_size, output_size)],
        ctypes.c_float(spatial_scale), ctypes.c_int(sampling_ratio), ctypes.c_int(int(aligned)), _p(out))
    if rc != 0:
        raise RuntimeError(f"oracle roi_align failed rc={rc}")
    return out


def msda_forward_numpy(value, shapes, lsi, loc, aw):
    """numpy convenience wrapper (same semantics)."""
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    return msda_forward(t(value), t(shapes), t(lsi), t(loc), t(aw)).numpy()
