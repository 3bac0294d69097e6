"""Depth backbone: the convolutional down-sampling path of DFormer
(ref models/dformer_backbone.py).  One-channel depth -> 128 channels at stride 16:
  stem   conv3x3/s2 (1->16) + BN + GELU + conv3x3/s2 (16->32) + BN
  then   BN + conv3x3/s2 (32->64),  BN + conv3x3/s2 (64->128),  [BN + conv3x3/s2 (128->256)]
The last stage exists (checkpoint keys) but the forward skips it (ref :142).  BatchNorm layers
are live nn.BatchNorm2d: running statistics in eval mode.
"""
import os
from typing import Dict, List

import torch
import torch.nn.functional as F
from torch import nn

from dfx import ops as _ops
from util.memo import memo_on
from util.misc import NestedTensor

from .position_encoding import build_position_encoding


class DownsamplePath(nn.Module):
    def __init__(self, in_channels: int, dims: List[int], train_backbone: bool = True,
                 freeze_batchnorm: bool = False):
        super().__init__()
        bn = lambda c: self._get_bn_layer(c, freeze_batchnorm)  # noqa: E731
        self.downsample_layers_e = nn.ModuleList([nn.Sequential(
            nn.Conv2d(in_channels, dims[0] // 2, kernel_size=3, stride=2, padding=1), bn(dims[0] // 2), nn.GELU(),
            nn.Conv2d(dims[0] // 2, dims[0], kernel_size=3, stride=2, padding=1), bn(dims[0]))])
        for a, b in zip(dims[:-1], dims[1:]):
            self.downsample_layers_e.append(
                nn.Sequential(bn(a), nn.Conv2d(a, b, kernel_size=3, stride=2, padding=1)))
        if not train_backbone:
            for p in self.downsample_layers_e.parameters():
                p.requires_grad = False

    @staticmethod
    def _get_bn_layer(num_features, freeze_batchnorm):
        layer = nn.BatchNorm2d(num_features)
        if freeze_batchnorm:
            layer.eval()
            for p in layer.parameters():
                p.requires_grad = False
        return layer


def _resize_mask(m, size):
    size = tuple(int(v) for v in size)
    return memo_on(m, ("resize_mask", size), lambda: F.interpolate(m[None].float(), size=size).to(torch.bool)[0])


class DFormerBackbone(nn.Module):
    def __init__(self, dims=(32, 64, 128, 256), train_backbone=True, return_interm_layers=False,
                 pretrained_path=None, freeze_batchnorm=True, eval_mode=False):
        super().__init__()
        self.return_interm_layers = return_interm_layers
        self.depth_backbone = DownsamplePath(1, list(dims), train_backbone, freeze_batchnorm)
        self._reset_parameters()
        self.strides = [2 ** (i + 2) for i in range(len(dims) - 1)]
        self.num_channels = dims[1:]
        if pretrained_path and not eval_mode:
            self.load_pretrained_weights(self.depth_backbone, pretrained_path)

    def _reset_parameters(self):
        for mod in self.modules():
            if isinstance(mod, nn.Conv2d):
                nn.init.xavier_uniform_(mod.weight)
                if mod.bias is not None:
                    nn.init.constant_(mod.bias, 0)
            elif isinstance(mod, nn.BatchNorm2d):
                nn.init.constant_(mod.weight, 1)
                nn.init.constant_(mod.bias, 0)

    def forward(self, tensor_list: NestedTensor) -> Dict[str, NestedTensor]:
        x, m = tensor_list.tensors, tensor_list.mask
        assert m is not None, "Input mask is None."
        out: Dict[str, NestedTensor] = {}
        stages = self.depth_backbone.downsample_layers_e[:-1]        # the 4th stage is never run
        if (not self.return_interm_layers and x.is_cuda and not torch.is_grad_enabled()
                and not any(mod.training for mod in stages.modules() if isinstance(mod, nn.BatchNorm2d))):
            x = self._forward_folded(stages, x)
            out["0"] = NestedTensor(x, _resize_mask(m, x.shape[-2:]))
            return out
        for i, stage in enumerate(stages):
            x = stage(x)
            if self.return_interm_layers:
                out[str(i)] = NestedTensor(x, _resize_mask(m, x.shape[-2:]))
        if not self.return_interm_layers:
            out["0"] = NestedTensor(x, _resize_mask(m, x.shape[-2:]))
        return out

    def _forward_folded(self, stages, x):
        """Inference with every BatchNorm (eval mode: running statistics) folded into the convolution BEFORE it:
        conv -> BN [-> BN of the next stage] is one convolution with scaled weights and a new bias, so the four
        normalisation passes over the full-resolution depth maps disappear.  A stage's leading BN follows the
        previous stage's last operation directly (only the last stage's output is returned here), and the zero
        padding of the next convolution is applied after it in both formulations, so the result is the same
        up to rounding."""
        key = tuple((t.data_ptr(), t._version) for mod in stages.modules() if isinstance(mod, (nn.Conv2d, nn.BatchNorm2d))
                    for t in list(mod.parameters(recurse=False)) + list(mod.buffers(recurse=False)))
        if getattr(self, "_folded", None) is None or self._folded[0] != key:
            ops = [mod for stage in stages for mod in stage]       # conv, bn, gelu, conv, bn | bn, conv | bn, conv
            plan = []                                              # (weight, bias, stride, padding, gelu_after)
            i = 0
            while i < len(ops):
                conv = ops[i]
                assert isinstance(conv, nn.Conv2d), "DownsamplePath starts every affine chain with a convolution"
                w = conv.weight.detach().clone()
                b = conv.bias.detach().clone() if conv.bias is not None else torch.zeros(w.shape[0], device=w.device)
                i += 1
                while i < len(ops) and isinstance(ops[i], nn.BatchNorm2d):
                    bn = ops[i]
                    scale = bn.weight.detach() * torch.rsqrt(bn.running_var + bn.eps)
                    w = w * scale.view(-1, 1, 1, 1)
                    b = (b - bn.running_mean) * scale + bn.bias.detach()
                    i += 1
                gelu = i < len(ops) and isinstance(ops[i], nn.GELU)
                i += int(gelu)
                # hand-written implicit-GEMM convolution, bias + GELU in its epilogue (csrc/conv_igemm.hip)
                plan.append(_ops.ConvPlan(w, b, conv.stride, conv.padding, conv.dilation, "gelu" if gelu else None,
                                          groups=conv.groups, padding_mode=conv.padding_mode))
            self._folded = (key, plan)
        for conv in self._folded[1]:
            x = conv(x)
        return x

    def load_pretrained_weights(self, model, pretrained_weights_path, prefix="downsample_layers_e"):
        """Partial load of a DFormer checkpoint ({'state_dict': ...}): conv / BN affine tensors whose
        key contains both ``prefix`` and the module's name; running statistics are left alone
        (ref :161-198)."""
        if not os.path.exists(pretrained_weights_path):
            print(f"Invalid path for pretrained weights: {pretrained_weights_path}")
            return
        weights = torch.load(pretrained_weights_path, map_location="cpu")["state_dict"]
        skip = ("running_mean", "running_var", "num_batches_tracked")
        for name, mod in model.named_modules():
            if not isinstance(mod, (nn.Conv2d, nn.BatchNorm2d)):
                continue
            for key, tensor in weights.items():
                if prefix in key and name in key and not any(s in key for s in skip):
                    if mod.weight.shape == tensor.shape:
                        mod.weight.data = tensor.data.clone()
                    if mod.bias is not None and "bias" in key:
                        mod.bias.data = tensor.data.clone()


class Joiner(nn.Sequential):
    def __init__(self, backbone, position_embedding):
        super().__init__(backbone, position_embedding)
        self.strides = backbone.strides
        self.num_channels = backbone.num_channels

    def forward(self, tensor_list: NestedTensor):
        xs = self[0](tensor_list)
        out = [x for _, x in sorted(xs.items())]
        pos = [self[1](x).to(x.tensors.dtype) for x in out]
        return out, pos


def build_dformer_backbone(args) -> Joiner:
    backbone = DFormerBackbone(dims=(32, 64, 128, 256), train_backbone=True, return_interm_layers=False,
                               pretrained_path=getattr(args, "dformer_weights", None), freeze_batchnorm=False,
                               eval_mode=False)
    return Joiner(backbone, build_position_encoding(args))
