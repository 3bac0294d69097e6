"""``RoIAlign`` module with the call surface of ``mmcv.ops.RoIAlign`` as the reference uses it
(/root/reference/models/deformable_transformer_multi_plusplus.py:129-132:
``RoIAlign(output_size=7, spatial_scale=1/32, sampling_ratio=2)``; mmcv-1.7.0 defaults
``pool_mode='avg'``, ``aligned=True``), on the gfx950 kernel behind include/dfx_roi.h.
"""
import torch
from torch import nn

from dfx import ops as _ops


class RoIAlign(nn.Module):
    def __init__(self, output_size, spatial_scale=1.0, sampling_ratio=0, pool_mode="avg", aligned=True,
                 use_torchvision=False):
        super().__init__()
        if pool_mode != "avg":
            raise NotImplementedError("only average pooling is used on this path")
        self.output_size = (output_size, output_size) if isinstance(output_size, int) else tuple(output_size)
        self.spatial_scale, self.sampling_ratio, self.aligned = float(spatial_scale), int(sampling_ratio), aligned

    def forward(self, input, rois):
        """input [N,C,H,W], rois [K,5] (batch_idx, x1, y1, x2, y2) -> [K,C,ph,pw]"""
        if input.dim() == 4 and input.stride(1) == 1 and input.shape[1] > 1:
            # channels-last memory (e.g. a permuted view of token-major encoder memory): no copy
            nhwc = input.permute(0, 2, 3, 1)
            if nhwc.is_contiguous():
                return self.forward_tokens(nhwc, rois).transpose(1, 2).reshape(
                    rois.shape[0], input.shape[1], *self.output_size)
        return _ops.roi_align(input.contiguous(), rois, self.output_size, self.spatial_scale, self.sampling_ratio,
                              self.aligned, channels_last=False)

    def forward_tokens(self, memory_nhwc, rois):
        """memory [N,H,W,C] -> [K, ph*pw, C] (the layout the query/RoI fusion head consumes)."""
        return _ops.roi_align(memory_nhwc, rois, self.output_size, self.spatial_scale, self.sampling_ratio,
                              self.aligned, channels_last=True)

    def __repr__(self):
        return (f"{self.__class__.__name__}(output_size={self.output_size}, spatial_scale={self.spatial_scale}, "
                f"sampling_ratio={self.sampling_ratio}, pool_mode=avg, aligned={self.aligned})")
