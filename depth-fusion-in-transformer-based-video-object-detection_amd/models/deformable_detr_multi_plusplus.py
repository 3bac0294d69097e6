"""TransVOD++ detector with RGB-D fusion (ref models/deformable_detr_multi_plusplus.py):
``DeformableDETR`` (:44-350), ``PostProcess`` (:550-582), ``build`` (:600-697).

The input batch is one clip: frame 0 is the current frame, frames 1..num_ref_frames the
reference frames; the output describes the CURRENT frame only.
"""
import copy

import torch
from torch import nn

from models.fused import Linear

from util.misc_multi import NestedTensor, inverse_sigmoid, nested_tensor_from_tensor_list

from .deformable_transformer_multi_plusplus import build_deforamble_transformer
from .detector_common import (MLP, DetectorBase, PostProcess, TrainingOnly, _prior_bias, _zero_last_layer,  # noqa: F401
                              apply_box_head, build_backbones, loss_weight_dict)


class DeformableDETR(DetectorBase):
    def __init__(self, backbone, depth_backbone, transformer, num_classes, num_queries, num_feature_levels,
                 num_ref_frames=3, aux_loss=True, with_box_refine=False, two_stage=False, use_depth=False,
                 depth_type=""):
        super().__init__()
        self.num_ref_frames = num_ref_frames
        self._init_common(backbone, depth_backbone, transformer, num_classes, num_queries, num_feature_levels,
                          aux_loss, with_box_refine, two_stage, use_depth, depth_type)
        hidden = transformer.d_model
        # heads of the temporal stage: one template, three independent copies (one per TQE/TDTD round)
        self.temp_class_embed = Linear(hidden, num_classes)
        self.temp_bbox_embed = MLP(hidden, hidden, 4, 3)
        self.temp_class_embed.bias.data = _prior_bias(num_classes)
        _zero_last_layer(self.temp_bbox_embed)
        nn.init.constant_(self.temp_bbox_embed.layers[-1].bias.data[2:], -2.0)
        self.temp_class_embed_list = nn.ModuleList([copy.deepcopy(self.temp_class_embed) for _ in range(3)])
        self.temp_bbox_embed_list = nn.ModuleList([copy.deepcopy(self.temp_bbox_embed) for _ in range(3)])
        self._finish_heads()

    def forward(self, samples: NestedTensor):
        if not isinstance(samples, NestedTensor):
            samples = nested_tensor_from_tensor_list(samples)
        _, _, h, w = samples.tensors.shape
        srcs, masks, pos, d_srcs, d_masks, d_pos, rgbd_query = self._encode_inputs(samples)
        query_embeds = None if self.two_stage else self.query_embed.weight
        res = self.transformer(srcs, masks, pos, d_srcs, d_masks, d_pos, (w, h, w, h), query_embeds,
                               self.class_embed[-1], self.bbox_embed[-1], self.temp_class_embed_list,
                               self.temp_bbox_embed_list, rgbd_query)
        if self.two_stage:      # the two-stage branch returns before the temporal stage (ref transformer :391-392)
            hs, init_reference, inter_references, enc_cls, enc_coord_unact = res
            final_hs, final_refs, out = None, None, {}
        else:
            hs, init_reference, inter_references, enc_cls, enc_coord_unact, final_hs, final_refs, out = res
        if self.two_stage:
            out["enc_outputs"] = {"pred_logits": enc_cls, "pred_boxes": enc_coord_unact.sigmoid()}
        if final_hs is not None:
            out["pred_logits"] = self.temp_class_embed_list[2](final_hs)
            out["pred_boxes"] = apply_box_head(self.temp_bbox_embed_list[2], final_hs, final_refs)
        return out

    @torch.jit.unused
    def _set_aux_loss(self, outputs_class, outputs_coord):
        return [{"pred_logits": a, "pred_boxes": b} for a, b in zip(outputs_class[:], outputs_coord[:])]


def build(args):
    if args.masks:
        raise NotImplementedError("the segmentation head is outside this path")
    backbone, depth_backbone = build_backbones(args)
    transformer = build_deforamble_transformer(args)
    model = DeformableDETR(backbone, depth_backbone, transformer, num_classes=args.num_classes,
                           num_queries=args.num_queries, num_feature_levels=args.num_feature_levels,
                           num_ref_frames=args.num_ref_frames, aux_loss=args.aux_loss,
                           with_box_refine=args.with_box_refine, two_stage=args.two_stage,
                           use_depth=args.use_depth, depth_type=args.depth_type)
    criterion = TrainingOnly(loss_weight_dict(args))
    return model, criterion, {"bbox": PostProcess()}
