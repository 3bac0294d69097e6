"""Operator wrappers with the reference's names and call signatures
(/root/reference/models/ops/functions/ms_deform_attn_func.py).

``MSDeformAttnFunction``        autograd op over the gfx950 library (ref :21-38)
``ms_deform_attn_core_pytorch`` pure-PyTorch debugging aid (ref :41-61), never on the product path
"""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

import MultiScaleDeformableAttention as MSDA


class MSDeformAttnFunction(Function):
    """apply(value, spatial_shapes, level_start_index, sampling_locations, attention_weights, im2col_step)"""

    @staticmethod
    def forward(ctx, value, value_spatial_shapes, value_level_start_index, sampling_locations,
                attention_weights, im2col_step):
        ctx.im2col_step = im2col_step
        ctx.save_for_backward(value, value_spatial_shapes, value_level_start_index, sampling_locations,
                              attention_weights)
        return MSDA.ms_deform_attn_forward(value, value_spatial_shapes, value_level_start_index,
                                           sampling_locations, attention_weights, im2col_step)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        saved = ctx.saved_tensors
        g_value, g_loc, g_aw = MSDA.ms_deform_attn_backward(*saved, grad_output.contiguous(), ctx.im2col_step)
        return g_value, None, None, g_loc, g_aw, None


def ms_deform_attn_core_pytorch(value, value_spatial_shapes, sampling_locations, attention_weights):
    """Plain-tensor statement of the operator for debugging: explicit corner gathers.

    value [N,S,M,D]; value_spatial_shapes iterable of (H,W); sampling_locations [N,Lq,M,L,P,2] in
    [0,1] (x,y); attention_weights [N,Lq,M,L,P] -> [N,Lq,M*D].  Same sampling rule as the kernels:
    pixel = loc*size - 0.5, bilinear, zeros outside the map.  Differentiable through autograd.
    """
    N, S, M, D = value.shape
    _, Lq, _, L, P, _ = sampling_locations.shape
    sizes = [(int(h), int(w)) for h, w in value_spatial_shapes]
    out = value.new_zeros(N, Lq, M, D)
    start = 0
    for lvl, (H, W) in enumerate(sizes):
        lvl_value = value[:, start:start + H * W]                       # [N,HW,M,D]
        start += H * W
        xy = sampling_locations[:, :, :, lvl]                            # [N,Lq,M,P,2]
        px = xy[..., 0] * W - 0.5
        py = xy[..., 1] * H - 0.5
        x0, y0 = torch.floor(px), torch.floor(py)
        fx, fy = px - x0, py - y0
        w_lvl = attention_weights[:, :, :, lvl]                          # [N,Lq,M,P]
        flat = lvl_value.permute(0, 2, 1, 3)                             # [N,M,HW,D]
        for dy, dx, wgt in ((0, 0, (1 - fy) * (1 - fx)), (0, 1, (1 - fy) * fx),
                            (1, 0, fy * (1 - fx)), (1, 1, fy * fx)):
            xi, yi = x0 + dx, y0 + dy
            ok = (xi >= 0) & (xi <= W - 1) & (yi >= 0) & (yi <= H - 1)
            idx = (yi.clamp(0, H - 1) * W + xi.clamp(0, W - 1)).long()   # [N,Lq,M,P]
            idx = idx.permute(0, 2, 1, 3).reshape(N, M, Lq * P, 1).expand(-1, -1, -1, D)
            got = torch.gather(flat, 2, idx).view(N, M, Lq, P, D).permute(0, 2, 1, 3, 4)
            coef = (wgt * ok.to(wgt.dtype) * w_lvl).unsqueeze(-1)        # [N,Lq,M,P,1]
            out = out + (got * coef).sum(3)
    return out.reshape(N, Lq, M * D)
