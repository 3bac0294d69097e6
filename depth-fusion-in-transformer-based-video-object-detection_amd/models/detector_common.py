"""Pieces shared by the three detector wrappers (the reference repeats them in
models/deformable_detr_{single,multi,multi_plusplus}.py): the prediction MLP, the input
projections, RGB / depth splitting and backbone dispatch, per-layer heads, and post-processing.
"""
import copy
import math

import torch
import torch.nn.functional as F
from torch import nn

from dfx import ops as _ops
from models.fused import Linear

from util import box_ops
from util.memo import memo_on
from util.misc import NestedTensor, inverse_sigmoid


def _get_clones(module, n):
    return nn.ModuleList([copy.deepcopy(module) for _ in range(n)])


class MLP(nn.Module):
    """Linear-ReLU stack ending in a plain Linear (ref deformable_detr_single.py:606-618)."""

    def __init__(self, input_dim, hidden_dim, output_dim, num_layers):
        super().__init__()
        self.num_layers = num_layers
        dims = [input_dim] + [hidden_dim] * (num_layers - 1) + [output_dim]
        self.layers = nn.ModuleList(Linear(a, b) for a, b in zip(dims[:-1], dims[1:]))

    def forward(self, x):
        fused = x.is_cuda and x.dtype == torch.float32 and not torch.is_grad_enabled() and x.shape[-1] % 4 == 0
        if fused:
            from dfx import ops as _ops            # bias + ReLU in the GEMM epilogue: one launch per layer
        for i, layer in enumerate(self.layers):
            last = i == self.num_layers - 1
            if fused and not last and layer.in_features % 4 == 0:
                x = _ops.linear(x.contiguous(), layer.weight, layer.bias, relu=True)
            else:
                x = layer(x)
                if not last:
                    x = F.relu(x)
        return x


class _ConvGN(nn.Sequential):
    """Conv2d + GroupNorm(32) of the input projections (ref deformable_detr_single.py:101-125).  GPU inference:
    the convolution on the hand-written kernels (1x1: MFMA GEMM with the bias in its epilogue; 3x3/2 of the extra
    levels: implicit GEMM), GroupNorm as two streaming launches that write token-major memory (dfx.ops.group_norm)."""

    def forward(self, x):
        conv, norm = self[0], self[1]
        if not (x.is_cuda and x.dtype == torch.float32 and not torch.is_grad_enabled()):
            return super().forward(x)
        x = x.contiguous()
        hw = x.shape[2] * x.shape[3]
        if conv.kernel_size == (1, 1) and conv.stride == (1, 1) and hw % 4 == 0 and x.shape[1] % 4 == 0:
            y = _ops.conv1x1(x, conv.weight, conv.bias)
        else:
            key = (conv.weight.data_ptr(), conv.weight._version, None if conv.bias is None else conv.bias._version)
            if getattr(self, "_plan", (None,))[0] != key:
                self._plan = (key, _ops.ConvPlan(conv.weight, conv.bias, conv.stride, conv.padding, conv.dilation,
                                                 groups=conv.groups, padding_mode=conv.padding_mode))
            y = self._plan[1](x)
        return _ops.group_norm(y, norm, tokens_out=True)


def _conv_gn(cin, cout, kernel_size=1, **kw):
    return _ConvGN(nn.Conv2d(cin, cout, kernel_size=kernel_size, **kw), nn.GroupNorm(32, cout))


def _prior_bias(num_classes, prior_prob=0.01):
    return torch.ones(num_classes) * (-math.log((1 - prior_prob) / prior_prob))


def _zero_last_layer(mlp):
    nn.init.constant_(mlp.layers[-1].weight.data, 0)
    nn.init.constant_(mlp.layers[-1].bias.data, 0)


def apply_box_head(bbox_embed, hs, reference):
    """sigmoid(MLP(hs) + inverse_sigmoid(reference)), reference being 2-d points or 4-d boxes."""
    box = bbox_embed(hs)
    if box.is_cuda and box.dtype == torch.float32 and not (torch.is_grad_enabled() and (box.requires_grad or reference.requires_grad)):
        from dfx import ops as _ops
        return _ops.box_refine(box, reference)
    unact = inverse_sigmoid(reference)
    if unact.shape[-1] == 4:
        box = box + unact
    else:
        assert unact.shape[-1] == 2
        box[..., :2] += unact
    return box.sigmoid()


class DetectorBase(nn.Module):
    """Backbones + projections + heads common to the single-frame, TransVOD and TransVOD++ detectors
    (ref deformable_detr_single.py:52-198, deformable_detr_multi_plusplus.py:44-204)."""

    NUM_CHANNELS = {"resnet18": [64, 128, 256, 512], "resnet50": [64, 512, 1024, 2048], "dformer": [32, 64, 128]}

    def _init_common(self, backbone, depth_backbone, transformer, num_classes, num_queries, num_feature_levels,
                     aux_loss, with_box_refine, two_stage, use_depth, depth_type):
        self.use_depth, self.depth_type = use_depth, depth_type
        self.crossfusion_features_concat = "crossfusion_2way_concat" in depth_type
        self.num_queries, self.aux_loss = num_queries, aux_loss
        self.with_box_refine, self.two_stage = with_box_refine, two_stage
        self.transformer = transformer
        hidden = transformer.d_model
        self.class_embed = Linear(hidden, num_classes)
        self.bbox_embed = MLP(hidden, hidden, 4, 3)
        self.num_feature_levels = num_feature_levels
        self.num_channels = dict(self.NUM_CHANNELS)
        if not two_stage:
            self.query_embed = nn.Embedding(num_queries, hidden * 2)
        self.input_proj = nn.ModuleList()
        if num_feature_levels > 1:
            n_out = len(backbone.strides)
            for i in range(n_out):
                self.input_proj.append(_conv_gn(self.num_channels[backbone.name][i], hidden))
            for _ in range(num_feature_levels - n_out):
                self.input_proj.append(_conv_gn(hidden, hidden, kernel_size=3, stride=2, padding=1))
        else:
            self.input_proj.append(_conv_gn(self.num_channels[backbone.name][-1], hidden))
        if use_depth:
            if self.crossfusion_features_concat:
                self.input_proj_depth = nn.ModuleList([_conv_gn(self.num_channels[backbone.d_name][-1], hidden)])
                self.concat_input_proj = nn.ModuleList([_conv_gn(hidden * 2, hidden)])
            elif "dformer" in depth_type and ("latefusion" in depth_type or "encoder_cf" in depth_type):
                self.input_proj_depth = nn.ModuleList([_conv_gn(128, hidden)])
        self.backbone, self.depth_backbone = backbone, depth_backbone

        self.class_embed.bias.data = _prior_bias(num_classes)
        _zero_last_layer(self.bbox_embed)
        nn.init.constant_(self.bbox_embed.layers[-1].bias.data[2:], -2.0)
        groups = [self.input_proj] + [getattr(self, n) for n in ("input_proj_depth", "concat_input_proj")
                                      if hasattr(self, n)]
        for group in groups:
            for proj in group:
                nn.init.xavier_uniform_(proj[0].weight, gain=1)
                nn.init.constant_(proj[0].bias, 0)

    def _finish_heads(self):
        """Per-decoder-layer heads; with box refinement they are independent copies and the decoder
        refines its reference boxes with them (ref deformable_detr_single.py:172-198)."""
        t = self.transformer
        num_pred = t.decoder.num_layers + 1 if self.two_stage else t.decoder.num_layers
        if self.with_box_refine:
            self.class_embed = _get_clones(self.class_embed, num_pred)
            self.bbox_embed = _get_clones(self.bbox_embed, num_pred)
            nn.init.constant_(self.bbox_embed[0].layers[-1].bias.data[2:], -2.0)
            t.decoder.bbox_embed = self.bbox_embed
        else:
            nn.init.constant_(self.bbox_embed.layers[-1].bias.data[2:], -2.0)
            self.class_embed = nn.ModuleList([self.class_embed for _ in range(num_pred)])
            self.bbox_embed = nn.ModuleList([self.bbox_embed for _ in range(num_pred)])
            t.decoder.bbox_embed = None
        if self.two_stage:
            t.decoder.class_embed = self.class_embed
            for box_embed in self.bbox_embed:
                nn.init.constant_(box_embed.layers[-1].bias.data[2:], 0.0)

    @staticmethod
    def with_pos_embed(tensor, pos):
        return tensor if pos is None else tensor + pos

    # ---- input side ---------------------------------------------------------------------------
    def _encode_inputs(self, samples: NestedTensor):
        """RGB/depth split, backbones, 1x1 projections -> (srcs, masks, pos, depth_srcs, depth_masks,
        depth_pos, rgbd_query)  (ref deformable_detr_single.py:227-322)."""
        x = samples.tensors
        if x.shape[1] == 4:
            assert self.use_depth, "Input tensors have 4 channels but use_depth is not set to True"
        if self.use_depth:
            assert x.shape[1] == 4, "Input tensors do not have 4 channels but use_depth is set to True"
        rgb, depth = samples, None
        if x.shape[1] == 4 and self.use_depth:
            rgb = NestedTensor(x[:, :3], samples.mask)
            depth = NestedTensor(x[:, 3:4], samples.mask)

        depth_features = depth_pos = None
        dt = self.depth_type
        if "crossfusion" in dt:
            features, pos, depth_features, depth_pos = self.backbone(samples)
        elif "latefusion" in dt or "encoder_cf" in dt:
            assert depth is not None, "Depth samples are None"
            features, pos, _, _ = self.backbone(rgb)
            depth_features, depth_pos = self.depth_backbone(depth)
        else:
            features, pos, _, _ = self.backbone(rgb)

        srcs, masks = [], []
        for l, feat in enumerate(features):
            src, mask = feat.decompose()
            assert mask is not None
            srcs.append(self.input_proj[l](src))
            masks.append(mask)
        for l in range(len(srcs), self.num_feature_levels):     # extra stride-2 levels
            src = self.input_proj[l](features[-1].tensors if l == len(features) else srcs[-1])
            size = tuple(int(v) for v in src.shape[-2:])
            mask = memo_on(samples.mask, ("resize_mask", size),
                           lambda: F.interpolate(samples.mask[None].float(), size=size).to(torch.bool)[0])
            srcs.append(src)
            masks.append(mask)
            pos.append(self.backbone[1](NestedTensor(src, mask)).to(src.dtype))

        depth_srcs, depth_masks = [], []
        if depth is not None and ("latefusion" in dt or "encoder_cf" in dt):
            for l, feat in enumerate(depth_features):
                src, mask = feat.decompose()
                assert mask is not None
                depth_srcs.append(self.input_proj_depth[l](src) if "dformer" in dt else src)
                depth_masks.append(mask)

        rgbd_query = []
        if self.crossfusion_features_concat:
            for l, feat in enumerate(depth_features):
                src, mask = feat.decompose()
                depth_srcs.append(self.input_proj_depth[l](src))
                depth_masks.append(mask)
            for i, (r, rp) in enumerate(zip(srcs, pos)):
                if r.shape[2:] != depth_srcs[i].shape[2:]:
                    depth_srcs[i] = F.interpolate(depth_srcs[i], size=r.shape[2:], mode="bilinear", align_corners=False)
                    depth_pos[i] = F.interpolate(depth_pos[i], size=r.shape[2:], mode="bilinear", align_corners=False)
                both = torch.cat([self.with_pos_embed(r, rp), self.with_pos_embed(depth_srcs[i], depth_pos[i])], dim=1)
                rgbd_query.append(self.concat_input_proj[0](both))
        return srcs, masks, pos, depth_srcs, depth_masks, depth_pos, rgbd_query

    @torch.jit.unused
    def _set_aux_loss(self, outputs_class, outputs_coord):
        return [{"pred_logits": a, "pred_boxes": b} for a, b in zip(outputs_class[:-1], outputs_coord[:-1])]


class PostProcess(nn.Module):
    """sigmoid scores -> top-100 over (query, class) -> box index = idx // C, label = idx % C ->
    xyxy boxes scaled to the image size (ref deformable_detr_single.py:569-603; its first top-k over
    all-but-the-last class is overwritten before use, so the effective rule is this one)."""

    num_select = 100

    @torch.no_grad()
    def forward(self, outputs, target_sizes):
        logits, boxes = outputs["pred_logits"], outputs["pred_boxes"]
        assert len(logits) == len(target_sizes) and target_sizes.shape[1] == 2
        n, _, c = logits.shape
        scores, idx = torch.topk(logits.sigmoid().view(n, -1), self.num_select, dim=1)
        box_idx, labels = idx // c, idx % c
        xyxy = torch.gather(box_ops.box_cxcywh_to_xyxy(boxes), 1, box_idx.unsqueeze(-1).repeat(1, 1, 4))
        img_h, img_w = target_sizes.unbind(1)
        xyxy = xyxy * torch.stack([img_w, img_h, img_w, img_h], dim=1)[:, None, :]
        return [{"scores": s, "labels": l, "boxes": b} for s, l, b in zip(scores, labels, xyxy)]


class TrainingOnly(nn.Module):
    """Placeholder for the reference's SetCriterion (Hungarian matching + focal/L1/GIoU losses).
    The training loss is outside the scope of this inference path."""

    def __init__(self, weight_dict=None):
        super().__init__()
        self.weight_dict = weight_dict or {}

    def forward(self, outputs, targets):
        raise NotImplementedError("the training criterion is not part of the MI355X inference path")


FUSION_TO_DEPTH_TYPE = {
    "Baseline": "Baseline_rgb",
    "LateFusion": "DepthDeform_latefusion_dformer",
    "Backbone_CrossFusion": "DepthDeform_dformer_crossfusion",
    "Encoder_CrossFusion": "DepthDeform_encoder_cf_dformer",
}


def build_backbones(args):
    """The --fusion_type -> depth_type translation and backbone choice shared by the three build()
    functions (ref deformable_detr_single.py:627-660)."""
    from .backbone_scratch import build_backbone_fromscratch
    from .dformer_backbone import build_dformer_backbone
    from .dformer_crossfusion_backbone import build_dformer_fusion_backbone
    if getattr(args, "dformer_weights", None):
        args.dformer_backbone = True
    try:
        args.depth_type = FUSION_TO_DEPTH_TYPE[args.fusion_type]
    except KeyError:
        raise NotImplementedError("Fusion type not implemented.")
    depth_backbone = None
    backbone = build_backbone_fromscratch(args)
    if "crossfusion" in args.depth_type:
        depth_backbone = build_dformer_fusion_backbone(args)   # stored, never called
    elif "latefusion" in args.depth_type or "encoder_cf" in args.depth_type:
        if getattr(args, "dformer_weights", None) or getattr(args, "dformer_backbone", False):
            depth_backbone = build_dformer_backbone(args)
        else:
            raise NotImplementedError("only the DFormer depth backbone is part of this path "
                                      "(the ResNet-18 depth backbone lives in the reference's research_scripts)")
    return backbone, depth_backbone


def loss_weight_dict(args):
    w = {"loss_ce": args.cls_loss_coef, "loss_bbox": args.bbox_loss_coef, "loss_giou": args.giou_loss_coef}
    if args.aux_loss:
        aux = {}
        for i in range(args.dec_layers - 1):
            aux.update({f"{k}_{i}": v for k, v in w.items()})
        aux.update({f"{k}_enc": v for k, v in w.items()})
        w.update(aux)
    return w
