"""TransVOD detector with RGB-D fusion (ref models/deformable_detr_multi.py): like the TransVOD++
wrapper but with a single pair of temporal heads (``temp_class_embed`` / ``temp_bbox_embed``) and a
transformer that takes only the last class head (:281).
"""
import torch
from torch import nn

from models.fused import Linear

from util.misc_multi import NestedTensor, nested_tensor_from_tensor_list

from .deformable_transformer_multi import build_deforamble_transformer
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
        self.temp_class_embed = Linear(hidden, num_classes)
        self.temp_bbox_embed = MLP(hidden, hidden, 4, 3)
        self.temp_class_embed.bias.data = _prior_bias(num_classes)
        _zero_last_layer(self.temp_bbox_embed)
        nn.init.constant_(self.temp_bbox_embed.layers[-1].bias.data[2:], -2.0)
        self._finish_heads()

    def forward(self, samples: NestedTensor):
        if not isinstance(samples, NestedTensor):
            samples = nested_tensor_from_tensor_list(samples)
        srcs, masks, pos, d_srcs, d_masks, d_pos, rgbd_query = self._encode_inputs(samples)
        query_embeds = None if self.two_stage else self.query_embed.weight
        res = self.transformer(srcs, masks, pos, d_srcs, d_masks, d_pos, query_embeds, self.class_embed[-1],
                               rgbd_query)
        out = {}
        if self.two_stage:
            out["enc_outputs"] = {"pred_logits": res[3], "pred_boxes": res[4].sigmoid()}
            return out
        final_hs, final_refs = res[5], res[6]
        if final_hs is not None:
            out["pred_logits"] = self.temp_class_embed(final_hs)
            out["pred_boxes"] = apply_box_head(self.temp_bbox_embed, final_hs, final_refs)
        return out


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
