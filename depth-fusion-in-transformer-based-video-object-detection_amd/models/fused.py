"""Switch for the GPU-only fused inference routes of the host modules.

Default module behaviour is the reference's op-by-op formulation (works on any device through
PyTorch).  ``enable_fused_inference(model)`` turns on, for every module that has one, the route
that runs on the hand-written gfx950 kernels (dfx.ops): folded frozen-BN + fused epilogues in the
ResNet, ... These routes have no CPU implementation and raise on CPU tensors; the MSDA operator
itself is always on the HIP kernels, fused route or not.
"""


def enable_fused_inference(model, enabled=True):
    n = 0
    for mod in model.modules():
        if hasattr(mod, "fused_inference"):
            mod.fused_inference = enabled
            n += 1
    return n


import torch
from torch import nn


class Linear(nn.Linear):
    """nn.Linear (same parameters, same state_dict keys) whose no-grad fp32 forward on the GPU is the hand-written
    MFMA GEMM with the bias in its epilogue (dfx.ops.linear, csrc/gemm_f32.hip); anything else is nn.Linear."""

    def forward(self, x):
        # (under autocast, or with parameters in another dtype / on another device, this is plain nn.Linear)
        if (x.is_cuda and x.dtype == torch.float32 and not torch.is_grad_enabled() and self.in_features % 4 == 0
                and x.numel() > 0 and self.weight.dtype == torch.float32 and self.weight.device == x.device
                and not torch.is_autocast_enabled()):
            from dfx import ops
            return ops.linear(x.contiguous(), self.weight, self.bias)
        return super().forward(x)


def post_is_fusable(post):
    """post = (residual, norm[, dropout]): may the add + LayerNorm ride in a GEMM epilogue?  Only when the sub-layer's
    Dropout is the identity (eval mode or p = 0) - the reference applies ``norm(residual + dropout(out))``
    (/root/reference/models/deformable_transformer_single.py:635-643) also under no_grad in train mode."""
    if post is None:
        return False
    drop = post[2] if len(post) > 2 else None
    return drop is None or not drop.training or drop.p == 0


def apply_post(post, out):
    """norm(residual + dropout(out)) for post = (residual, norm[, dropout]); ``out`` itself when post is None."""
    if post is None:
        return out
    drop = post[2] if len(post) > 2 else None
    if drop is not None:
        out = drop(out)
    return post[1](post[0] + out)
