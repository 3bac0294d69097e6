"""Backbone Cross Fusion (ref models/dformer_crossfusion_backbone.py).

What runs in the reference and is reproduced here is the fusion BLOCK: ``fuse_layers`` (:387-428)
- flatten two feature maps to tokens, build the reference grid at the SOURCE resolution scaled by
the TARGET's valid ratios, run one ``DepthDeformableTransformerEncoderLayer`` (:120-181: MSDA
cross-attention source <- target + Linear/LayerNorm epilogues), unflatten.

The surrounding ``FusionBackboneBase.forward`` is not reachable from ``build_model``: ``build()``
stores this backbone in ``depth_backbone`` and the detector only ever calls ``self.backbone`` for
the "crossfusion" type (deformable_detr_single.py:249-251, 649-654), and the forward itself fails
on its own channel counts (projections built for 256/512/1024 channels, applied to 512/1024/2048;
SURVEY.md section 0.7).  ``FusionBackbone`` here therefore owns the same parameters under the same
names (checkpoints with ``depth_backbone.0.*`` keys load) and raises if its forward is called.
"""
from typing import List

import torch
from torch import nn

from models.fused import Linear

from models.ops.modules import MSDeformAttn
from util.misc import NestedTensor

from .backbone_scratch import FrozenBatchNorm2d
from .dformer_backbone import DownsamplePath
from .position_encoding import build_position_encoding
from .resnet import ResNet50
from .transformer_layers import (_add_pos, _get_activation_fn, get_reference_points, get_valid_ratio,
                                 make_level_tensors)


class DepthDeformableTransformerEncoderLayer(nn.Module):
    """Fusion layer of the backbone variant: like the Late Fusion layer but with a configurable FFN
    activation (ReLU by default, ref :120-181)."""

    def __init__(self, d_model=256, d_ffn=1024, dropout=0.1, activation="relu", n_depth_levels=1, n_heads=8,
                 dpth_n_points=4, depth_self_attn=True):
        super().__init__()
        self.depth_self_attn = depth_self_attn
        self.cross_attn = MSDeformAttn(d_model, n_depth_levels, n_heads, dpth_n_points)
        self.dropout1 = nn.Dropout(dropout)
        self.norm1 = nn.LayerNorm(d_model)
        self.linear1 = Linear(d_model, d_model)
        self.activation = _get_activation_fn(activation)
        self.dropout4 = nn.Dropout(dropout)
        self.norm3 = nn.LayerNorm(d_model)
        self.depth_scale_adapt = Linear(d_model, d_model)
        self.norm_depth_scale = nn.LayerNorm(d_model)
        self.cross_scale_adapt = Linear(d_model, d_model)

    with_pos_embed = staticmethod(_add_pos)

    def forward_ffn(self, tgt):
        return self.norm3(tgt + self.dropout4(self.activation(self.linear1(tgt))))

    def forward(self, tgt, query_pos, src_pos, tgt_spatial_shapes, reference_points, depth_reference_points, src,
                src_spatial_shapes, frame_start_index, tgt_padding_mask=None, src_padding_mask=None):
        src = self.norm_depth_scale(self.depth_scale_adapt(src))
        y = self.cross_attn(_add_pos(tgt, query_pos), reference_points, src, src_spatial_shapes,
                            frame_start_index, src_padding_mask)
        tgt = self.norm1(tgt + self.dropout1(self.cross_scale_adapt(y)))
        return self.forward_ffn(tgt)


def fuse_layers(src, target, pos_src, pos_target, mask_src, mask_target, fusion_layer):
    """src/target [N,C,h,w] maps (already projected to d_model), pos_* their positional embeddings,
    mask_* [N,h,w] padding masks.  Returns the fused source map, same shape as ``src``."""
    tok = lambda x: x.flatten(2).transpose(1, 2)  # noqa: E731
    shapes_src, _ = make_level_tensors([src.shape[-2:]], src.device)
    shapes_tgt, lsi_tgt = make_level_tensors([target.shape[-2:]], target.device)
    ratios_src = torch.stack([get_valid_ratio(mask_src)], dim=1)
    ratios_tgt = torch.stack([get_valid_ratio(mask_target)], dim=1)
    ref_for_src_queries = get_reference_points(shapes_src, ratios_tgt, target.device)
    ref_unused = None    # the reference also builds a grid at the target resolution; the layer ignores it
    del ratios_src
    fused = fusion_layer(tok(src), tok(pos_src), tok(pos_target), shapes_src, ref_for_src_queries, ref_unused,
                         tok(target), shapes_tgt, lsi_tgt, mask_src.flatten(1), mask_target.flatten(1))
    return fused.transpose(1, 2).reshape(src.shape)


class FusionBackboneBase(nn.Module):
    get_valid_ratio = staticmethod(get_valid_ratio)
    get_reference_points = staticmethod(get_reference_points)
    fuse_layers = staticmethod(fuse_layers)

    def __init__(self, rgb_name, d_name, rgb_backbone, depth_backbone, position_embedding, train_backbone,
                 return_interm_layers, fusion_mode, fusion_layers: List[int], d_model, bidirectional,
                 dim_feedforward=1024, dropout=0.1, activation="relu", n_head=8, fusion_levels=1,
                 fusion_n_points=4, depth_pretrained_path=None, eval=False):
        super().__init__()
        assert rgb_name in ["resnet50"] and d_name in ["dformer"], "Fusion Backbone not implemented"
        if not train_backbone:
            for p in list(rgb_backbone.parameters()) + list(depth_backbone.parameters()):
                p.requires_grad = False
        self.name = self.rgb_name = rgb_name
        self.d_name = d_name
        self.body, self.d_body = rgb_backbone, depth_backbone
        self.position_embedding = position_embedding
        self.fusion_mode, self.fusion_layers = fusion_mode, fusion_layers
        self.return_interm_layers, self.d_model, self.bidirectional = return_interm_layers, d_model, bidirectional
        self.depth_self_attn = True
        self.model_strides = {"resnet18": [2, 8, 16, 32], "resnet50": [2, 4, 16, 32], "dformer": [4, 8, 16]}
        self.model_num_channels = {"resnet18": [64, 128, 256, 512], "resnet50": [256, 512, 1024, 2048],
                                   "dformer": [32, 64, 128, 256]}
        self.return_layer_no = [3, 4] if return_interm_layers else [4]
        self.strides = [8, 16, 32] if return_interm_layers else [32]
        self.depth_strides = [4, 8, 16]
        self.depth_num_channels = [32, 64, 128, 256]
        self.return_layers = {f"layer{i}": str(k) for k, i in enumerate(self.return_layer_no)}
        assert not any(l in self.fusion_layers for l in (0, 1)), \
            "Fusion layers 0 and 1 not supported as dformer has only 2 levels"
        for layer in self.fusion_layers:
            if layer not in (2, 3, 4):
                continue
            rgb_c = self.model_num_channels[rgb_name][layer - 2]
            d_c = self.model_num_channels[d_name][layer - 2]
            gn_d = {2: 4, 3: 8, 4: 16}[layer]
            proj = lambda cin, cout, g: nn.Sequential(nn.Conv2d(cin, cout, kernel_size=1), nn.GroupNorm(g, cout))  # noqa: E731
            setattr(self, f"input_rgb_proj{layer}", proj(rgb_c, d_model, 32))
            setattr(self, f"output_rgb_proj{layer}", proj(d_model, rgb_c, 32))
            setattr(self, f"input_d_proj{layer}", proj(d_c, d_model, gn_d))
            setattr(self, f"output_d_proj{layer}", proj(d_model, d_c, gn_d))
            mk = lambda: DepthDeformableTransformerEncoderLayer(  # noqa: E731
                d_model, dim_feedforward, dropout, activation, fusion_levels, n_head, fusion_n_points,
                depth_self_attn=self.depth_self_attn)
            setattr(self, f"d2r_fusion{layer}", mk())
            if self.bidirectional:
                setattr(self, f"r2d_fusion{layer}", mk())
        self._reset_parameters()

    def _reset_parameters(self):
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        for m in self.modules():
            if isinstance(m, MSDeformAttn):
                m._reset_parameters()
            elif isinstance(m, nn.Conv2d):
                nn.init.xavier_uniform_(m.weight)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, (nn.BatchNorm2d, nn.LayerNorm)):
                nn.init.constant_(m.weight, 1.0)
                nn.init.constant_(m.bias, 0.0)

    def forward(self, tensor_list: NestedTensor):
        raise NotImplementedError(
            "FusionBackboneBase.forward is unreachable from build_model in the reference and inconsistent "
            "with its own projection sizes (SURVEY.md 0.7); use fuse_layers() with a d2r_fusion{k} layer.")


class FusionBackbone(FusionBackboneBase):
    def __init__(self, rgb_name, d_name, train_backbone, position_embedding, return_interm_layers, dilation,
                 depth_type, fusion_layers, d_model, bidirectional, depth_pretrained_path=None, eval=False):
        if d_name != "dformer":
            raise NotImplementedError(f"Backbone {d_name} not implemented")
        body = ResNet50(FrozenBatchNorm2d, replace_stride_with_dilation=[False, False, dilation])
        d_body = DownsamplePath(in_channels=1, dims=[32, 64, 128, 256], train_backbone=True, freeze_batchnorm=False)
        super().__init__(rgb_name, d_name, body, d_body, position_embedding, train_backbone, return_interm_layers,
                         fusion_mode=depth_type, fusion_layers=fusion_layers, d_model=d_model,
                         bidirectional=bidirectional, depth_pretrained_path=depth_pretrained_path, eval=eval)
        if dilation:
            self.strides[-1] = self.strides[-1] // 2


class Joiner(nn.Sequential):
    def __init__(self, backbone, position_embedding):
        super().__init__(backbone, position_embedding)
        self.strides = backbone.strides
        self.num_channels = backbone.model_num_channels
        self.rgb_name, self.d_name = backbone.rgb_name, backbone.d_name

    def forward(self, tensor_list: NestedTensor):
        xs, xd = self[0](tensor_list)
        out = [x for _, x in sorted(xs.items())]
        d_out = [x for _, x in sorted(xd.items())]
        return (out, [self[1](x).to(x.tensors.dtype) for x in out],
                d_out, [self[1](x).to(x.tensors.dtype) for x in d_out])


def build_dformer_fusion_backbone(args):
    position_embedding = build_position_encoding(args)
    fusion_layers = [2, 3, 4] if "crossfusion" in args.depth_type else [4]
    backbone = FusionBackbone(
        rgb_name=args.backbone, d_name="dformer", train_backbone=args.lr_backbone > 0,
        position_embedding=position_embedding,
        return_interm_layers=args.masks or (args.num_feature_levels > 1), dilation=args.dilation,
        depth_type=args.depth_type, fusion_layers=fusion_layers, d_model=256,
        bidirectional="crossfusion_2way" in args.depth_type,
        depth_pretrained_path=getattr(args, "dformer_weights", False), eval=False)
    return Joiner(backbone, position_embedding)
