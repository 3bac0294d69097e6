"""Drop-in for the reference's compiled extension module of the same name.

The reference builds ``MultiScaleDeformableAttention`` from models/ops/src with
pybind11 (/root/reference/models/ops/src/vision.cpp:13-16, setup.py:51-59) and imports
it at module import time (models/ops/functions/ms_deform_attn_func.py:18).  This module
exposes the same two functions with the same signatures, backed by the gfx950 C-ABI
library (include/dfx_msda.h) instead of the CUDA extension.  With this directory on
``sys.path`` the reference's own ``ms_deform_attn_func.py`` imports it unchanged.
"""
from dfx.ops import msda_backward as _bwd
from dfx.ops import msda_forward as _fwd


def ms_deform_attn_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step):
    """value [N,S,M,D], spatial_shapes i64 [L,2], level_start_index i64 [L],
    sampling_loc [N,Lq,M,L,P,2], attn_weight [N,Lq,M,L,P] -> [N,Lq,M*D]
    (ms_deform_attn.h:20-38)."""
    return _fwd(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step)


def ms_deform_attn_backward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output,
                            im2col_step):
    """-> [grad_value, grad_sampling_loc, grad_attn_weight] (ms_deform_attn.h:41-61)."""
    return _bwd(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output, im2col_step)
