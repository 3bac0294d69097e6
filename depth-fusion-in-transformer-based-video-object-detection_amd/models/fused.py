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
