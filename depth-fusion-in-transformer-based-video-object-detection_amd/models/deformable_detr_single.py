"""Single-frame Deformable-DETR with RGB-D fusion (ref models/deformable_detr_single.py):
``DeformableDETR`` (:31-370), ``PostProcess`` (:569-603), ``MLP`` (:606-618), ``build`` (:621-709).
"""
import torch
from torch import nn

from util.misc import NestedTensor, inverse_sigmoid, nested_tensor_from_tensor_list

from .deformable_transformer_single import build_deforamble_transformer
from .detector_common import (MLP, DetectorBase, PostProcess, TrainingOnly, apply_box_head,  # noqa: F401
                              build_backbones, loss_weight_dict)


class DeformableDETR(DetectorBase):
    """samples: NestedTensor([B,3|4,H,W], mask[B,H,W]) or a list of [C,H,W] images
    -> {"pred_logits" [B,Q,classes], "pred_boxes" [B,Q,4] (cx,cy,w,h in [0,1]), "aux_outputs"}."""

    def __init__(self, backbone, depth_backbone, transformer, num_classes, num_queries, num_feature_levels,
                 aux_loss=True, with_box_refine=False, two_stage=False, use_depth=False, depth_type="Baseline_rgb"):
        super().__init__()
        self._init_common(backbone, depth_backbone, transformer, num_classes, num_queries, num_feature_levels,
                          aux_loss, with_box_refine, two_stage, use_depth, depth_type)
        self._finish_heads()

    def forward(self, samples: NestedTensor):
        if not isinstance(samples, NestedTensor):
            samples = nested_tensor_from_tensor_list(samples)
        srcs, masks, pos, d_srcs, d_masks, d_pos, rgbd_query = self._encode_inputs(samples)
        query_embeds = None if self.two_stage else self.query_embed.weight
        hs, init_reference, inter_references, enc_cls, enc_coord_unact = self.transformer(
            srcs, masks, pos, d_srcs, d_masks, d_pos, query_embeds, rgbd_query)
        classes, coords = [], []
        for lvl in range(hs.shape[0]):
            reference = init_reference if lvl == 0 else inter_references[lvl - 1]
            classes.append(self.class_embed[lvl](hs[lvl]))
            coords.append(apply_box_head(self.bbox_embed[lvl], hs[lvl], reference))
        classes, coords = torch.stack(classes), torch.stack(coords)
        out = {"pred_logits": classes[-1], "pred_boxes": coords[-1]}
        if self.aux_loss:
            out["aux_outputs"] = self._set_aux_loss(classes, coords)
        if self.two_stage:
            out["enc_outputs"] = {"pred_logits": enc_cls, "pred_boxes": enc_coord_unact.sigmoid()}
        return out


def build(args):
    num_classes = args.num_classes
    if args.masks:
        raise NotImplementedError("the segmentation head is outside this path")
    backbone, depth_backbone = build_backbones(args)
    transformer = build_deforamble_transformer(args)
    model = DeformableDETR(backbone, depth_backbone, transformer, num_classes=num_classes,
                           num_queries=args.num_queries, num_feature_levels=args.num_feature_levels,
                           aux_loss=args.aux_loss, with_box_refine=args.with_box_refine, two_stage=args.two_stage,
                           use_depth=args.use_depth, depth_type=args.depth_type)
    criterion = TrainingOnly(loss_weight_dict(args))
    return model, criterion, {"bbox": PostProcess()}
