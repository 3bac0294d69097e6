"""RGB backbone: ResNet-50 with frozen batch-norm (+ DC5 dilation) and the positional-encoding
joiner (ref models/backbone_scratch.py).  Reference symbols kept: ``FrozenBatchNorm2d`` (:31-68),
``FusionBackboneBase`` / ``FusionBackbone`` (:71-165), ``Joiner`` (:168-187),
``build_backbone_fromscratch`` (:190-201).

One reference quirk is kept on purpose: with ``return_interm_layers`` the output
dict key "0" written for layer2 is overwritten by layer4, so the returned order is
[layer4, layer2, layer3].
"""
from typing import Dict, List

import torch
import torch.nn.functional as F
from torch import nn

from util.memo import memo_on
from util.misc import NestedTensor

from .position_encoding import build_position_encoding
from .resnet import ResNet50


class FrozenBatchNorm2d(nn.Module):
    """y = x * w/sqrt(var+eps) + (b - mean*w/sqrt(var+eps)) with all four statistics as buffers."""

    def __init__(self, n, eps=1e-5):
        super().__init__()
        self.register_buffer("weight", torch.ones(n))
        self.register_buffer("bias", torch.zeros(n))
        self.register_buffer("running_mean", torch.zeros(n))
        self.register_buffer("running_var", torch.ones(n))
        self.eps = eps

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys,
                              unexpected_keys, error_msgs):
        state_dict.pop(prefix + "num_batches_tracked", None)
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys,
                                      unexpected_keys, error_msgs)

    def scale_shift(self):
        scale = self.weight * (self.running_var + self.eps).rsqrt()
        return scale, self.bias - self.running_mean * scale

    def forward(self, x):
        scale, shift = self.scale_shift()
        return x * scale.reshape(1, -1, 1, 1) + shift.reshape(1, -1, 1, 1)


def _resize_mask(m, size):
    size = tuple(int(v) for v in size)
    return memo_on(m, ("resize_mask", size), lambda: F.interpolate(m[None].float(), size=size).to(torch.bool)[0])


class FusionBackboneBase(nn.Module):
    def __init__(self, rgb_name: str, d_name: str, rgb_backbone: nn.Module, depth_backbone, position_embedding,
                 train_backbone: bool, return_interm_layers: bool, fusion_mode: str, fusion_layers: List[int],
                 d_model: int, bidirectional: bool, **_unused):
        super().__init__()
        assert rgb_name in ["resnet50"], f"Backbone {rgb_name} not supported"
        if not train_backbone:
            for p in rgb_backbone.parameters():
                p.requires_grad = False
        self.name = self.rgb_name = rgb_name
        self.d_name = d_name
        self.body = rgb_backbone
        self.position_embedding = position_embedding
        self.fusion_mode, self.fusion_layers = fusion_mode, fusion_layers
        self.return_interm_layers = return_interm_layers
        self.d_model, self.bidirectional = d_model, bidirectional
        self.model_strides = {"resnet18": [2, 8, 16, 32], "resnet50": [2, 4, 16, 32]}
        self.model_num_channels = {"resnet18": [64, 128, 256, 512], "resnet50": [256, 512, 1024, 2048]}
        self.return_layer_no = [2, 3, 4] if return_interm_layers else [4]
        self.strides = [8, 16, 32] if return_interm_layers else [32]
        self.return_layers = {f"layer{i}": str(k) for k, i in enumerate(self.return_layer_no)}
        # set by models.fused.enable_fused_inference(): GPU-only inference path with frozen BN folded
        # into the convolutions and fused bias/residual/ReLU epilogues
        self.fused_inference = False

    def forward(self, tensor_list: NestedTensor):
        x = tensor_list.tensors[:, :3]
        m = tensor_list.mask
        assert m is not None, "Mask should not be None"
        body = self.body
        fused = self.fused_inference and not torch.is_grad_enabled()
        own = hasattr(body, "stem")      # models.resnet.ResNet50; any torchvision-style body works op by op (ref :112-131)
        x = body.stem(x, fused) if own else body.maxpool(body.relu(body.bn1(body.conv1(x))))
        out: Dict[str, NestedTensor] = {}
        wanted = set(self.return_layers.values()) if self.return_interm_layers else set()
        for key, stage in (("0", body.layer1), ("1", body.layer2), ("2", body.layer3), ("3", body.layer4)):
            x = body.run_stage(stage, x, fused) if own else stage(x)
            if key == "3":
                # the last stage always reports; under key "3" only if such a key was requested
                out["3" if key in wanted else "0"] = NestedTensor(x, _resize_mask(m, x.shape[-2:]))
            elif key in wanted:
                out[key] = NestedTensor(x, _resize_mask(m, x.shape[-2:]))
        return out, None


class FusionBackbone(FusionBackboneBase):
    """ResNet backbone with frozen BatchNorm."""

    def __init__(self, rgb_name, d_name, train_backbone, position_embedding, return_interm_layers, dilation,
                 depth_type, fusion_layers, d_model, bidirectional):
        assert rgb_name == "resnet50", f"Backbone {rgb_name} not supported"
        # NOTE: the reference asks torchvision for ImageNet weights here (pretrained=is_main_process());
        # there is no network on this stack - weights come from the checkpoint the caller loads.
        body = ResNet50(FrozenBatchNorm2d, replace_stride_with_dilation=[False, False, dilation])
        super().__init__(rgb_name, d_name, body, None, position_embedding, train_backbone, return_interm_layers,
                         fusion_mode=depth_type, fusion_layers=fusion_layers, d_model=d_model,
                         bidirectional=bidirectional)
        if dilation:
            self.strides[-1] = self.strides[-1] // 2


class Joiner(nn.Sequential):
    def __init__(self, backbone, position_embedding):
        super().__init__(backbone, position_embedding)
        self.strides = backbone.strides
        self.num_channels = backbone.model_num_channels
        self.name = self.rgb_name = backbone.name
        self.d_name = backbone.d_name

    def forward(self, tensor_list: NestedTensor):
        xs, _ = self[0](tensor_list)
        out = [x for _, x in sorted(xs.items())]
        pos = [self[1](x).to(x.tensors.dtype) for x in out]
        return out, pos, None, None


def build_backbone_fromscratch(args):
    position_embedding = build_position_encoding(args)
    train_backbone = args.lr_backbone > 0
    return_interm_layers = args.masks or (args.num_feature_levels > 1)
    fusion_layers = [1, 2, 3] if "crossfusion" in args.depth_type else [3]
    bidirectional = "2way" in args.depth_type
    backbone = FusionBackbone(args.backbone, "resnet18", train_backbone, position_embedding, return_interm_layers,
                              args.dilation, args.depth_type, fusion_layers, 256, bidirectional)
    return Joiner(backbone, position_embedding)
