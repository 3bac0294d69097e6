"""TransVOD++ transformer with RGB-D fusion (ref models/deformable_transformer_multi_plusplus.py).

The batch axis of the inputs is the clip: frame 0 is the current frame, frames 1..R are reference
frames.  The spatial stage (Late Fusion / encoder / decoder) runs on all T = 1+R frames and is
shared with the single-frame model (``SpatialTransformerBase``).  The temporal stage (ref :401-601)

  per frame   class logits and boxes from the last decoder layer, 300 RoIs on the frame's encoder
              memory (current frame: plain memory; reference frames: memory + positional
              embedding), RoIAlign 7x7, query/RoI fusion ``dynamic_layer_for_current_query1``
  per clip    3 x [ top-(k*R) reference queries by class-1 score, k = 80/50/30 -> gather ->
              ``temporal_query_layer{i}`` (TQE) -> ``temporal_decoder{i}`` (TDTD) on the current memory ]

is written here as two functions, ``frame_stage`` and ``temporal_stage``, so that a clip whose
frames are spread over several GPUs can exchange just the per-frame results (models/clip_inference.py).
``forward`` composes them exactly as the reference does for one clip on one device.

Quirks of the reference that are kept (SURVEY.md sections 0.6 and 8a, row a11): the temporal
decoders get [1,300,R,4] reference boxes against a 1-level value map (flat read of the location
tensor); RoIAlign runs with spatial_scale 1/32 on the stride-16 map; the (h, w) used to view the
memory as a map are those of the last depth level under Late Fusion; only
``dynamic_layer_for_current_query1`` is used, ``...query2/3`` exist for checkpoint compatibility.
"""
import torch
from torch import nn

from util import box_ops
from util.misc import inverse_sigmoid

from .deformable_transformer_single import SpatialTransformerBase
from .detector_common import apply_box_head
from .roi_align import RoIAlign
from .sparse_roi_head.head import RCNNHead
from .transformer_layers import (DeformableTransformerDecoder, DeformableTransformerDecoderLayer,  # noqa: F401
                                 DeformableTransformerEncoder, DeformableTransformerEncoderLayer,
                                 DeformableTransformerFusionLayerV2, DepthDeformableTransformerEncoderLayer,
                                 RGBDDeformableTransformerEncoderV2, TemporalDeformableTransformerDecoder,
                                 TemporalDeformableTransformerEncoderLayer, TemporalQueryEncoder,
                                 TemporalQueryEncoderLayer, _get_activation_fn, _get_clones,
                                 get_reference_points, get_valid_ratio)

TOPK_PER_REF = (80, 50, 30)


def get_box_tensor(boxes):
    return boxes.tensor if hasattr(boxes, "tensor") else boxes


def bbox2roi(bbox_list):
    """list of [n_i,4] boxes (one entry per image) -> [sum n_i, 5] rows (image index, x1, y1, x2, y2)."""
    rows = []
    for img_id, boxes in enumerate(bbox_list):
        boxes = get_box_tensor(boxes)
        rows.append(torch.cat([boxes.new_full((boxes.size(0), 1), img_id), boxes], dim=-1))
    return torch.cat(rows, 0)


class DeformableTransformer(SpatialTransformerBase):
    def __init__(self, d_model=256, nhead=8, num_encoder_layers=6, num_decoder_layers=6, dim_feedforward=1024,
                 dropout=0.1, activation="relu", return_intermediate_dec=False, num_feature_levels=4,
                 dec_n_points=4, enc_n_points=4, two_stage=False, two_stage_num_proposals=300, num_query=300,
                 n_temporal_decoder_layers=1, num_ref_frames=3, fixed_pretrained_model=False, args=None,
                 use_depth=False, depth_type="", dpth_feature_levels=1, dpth_n_points=4):
        super().__init__()
        self.num_ref_frames = num_ref_frames
        self.fixed_pretrained_model = fixed_pretrained_model
        self.n_temporal_query_layers = 3
        self.num_query = num_query
        dec_layer = self._build_spatial(d_model, nhead, num_encoder_layers, num_decoder_layers, dim_feedforward,
                                        dropout, activation, return_intermediate_dec, num_feature_levels,
                                        dec_n_points, enc_n_points, two_stage, two_stage_num_proposals, use_depth,
                                        depth_type, dpth_feature_levels, dpth_n_points)
        self.temporal_roi_layers1 = nn.ModuleList(
            [RoIAlign(output_size=7, spatial_scale=1 / s, sampling_ratio=2) for s in [32]])
        self.cfg = {"MODEL": {"SparseRCNN": {"NHEADS": 8, "DROPOUT": 0.0, "DIM_FEEDFORWARD": 2048,
                                              "ACTIVATION": "relu", "HIDDEN_DIM": d_model, "NUM_CLS": 1,
                                              "NUM_REG": 3, "NUM_HEADS": 6, "NUM_DYNAMIC": 2, "DIM_DYNAMIC": 64},
                              "ROI_BOX_HEAD": {"POOLER_RESOLUTION": 7}}}
        for i in (1, 2, 3):
            setattr(self, f"temporal_query_layer{i}",
                    TemporalQueryEncoderLayer(d_model, dim_feedforward, dropout, activation, nhead))
        for i in (1, 2, 3):
            setattr(self, f"dynamic_layer_for_current_query{i}",
                    RCNNHead(self.cfg, d_model, 3, dim_feedforward, nhead, dropout, activation))
        for i in (1, 2, 3):
            setattr(self, f"temporal_decoder{i}",
                    TemporalDeformableTransformerDecoder(dec_layer, n_temporal_decoder_layers, False))
        self._reset_parameters()

    # ------------------------------------------------------------------------------------------
    def frame_stage(self, hs_last, ref_last, memory, pos_embed, hw, imgs_whwh, class_embed, bbox_embed,
                    roles=("cur", "ref")):
        """Per-frame half of the temporal stage for F frames (any subset of a clip), batched.

        hs_last [F,Q,C] last decoder layer output, ref_last [F,Q,4] its reference boxes,
        memory [F,S,C], pos_embed [F,S,C] (level-embedded positional embedding), hw = (h, w).
        Returns a dict with, per frame:
          logits [F,Q,classes]; boxes [F,Q,4] (sigmoid, cxcywh)
          "cur": queries fused with RoI features of the plain memory              [F,Q,C]
          "ref": queries fused with RoI features of memory + positional embedding [F,Q,C]
        (ref :450-518; a frame needs "cur" when it is the current frame and "ref" when it serves
        as a reference frame.  The reference runs RoIAlign and the fusion head once per frame; the
        head's self-attention is per batch element, so all frames go through in one call.)"""
        h, w = hw
        F_, Q, C = hs_last.shape
        logits = class_embed(hs_last)
        boxes = apply_box_head(bbox_embed, hs_last, ref_last)
        xyxy = box_ops.box_cxcywh_to_xyxy(boxes) * imgs_whwh                      # image pixels
        rois = bbox2roi([xyxy[f] for f in range(F_)])                             # [F*Q,5], image index = frame
        out = dict(logits=logits, boxes=boxes)
        roi = self.temporal_roi_layers1[0]
        head = self.dynamic_layer_for_current_query1
        pooled = []
        for role in roles:
            mem = memory if role == "cur" else memory + pos_embed
            fmap = mem.permute(0, 2, 1).unsqueeze(-1).view(F_, C, h, w)           # channels-last view, no copy
            pooled.append(roi(fmap, rois))
        # both roles share the head's query self-attention and its dynamic_layer (they see hs_last only): one pass
        fused = head(pooled, hs_last) if len(pooled) > 1 else [head(pooled[0], hs_last)]
        for role, y in zip(roles, fused):
            out[role] = y.view(F_, Q, C)
        return out

    def temporal_stage(self, cur_hs, cur_reference_out, cur_memory, ref_pool, logit_pool, others,
                       spatial_shapes, level_start_index, valid_ratio_cur, temp_class_embed_list,
                       temp_bbox_embed_list):
        """Per-clip half for F current frames at once (ref :525-601 runs it for one).

        cur_hs [F,Q,C] ("cur" fused queries), cur_reference_out [F,Q,4], cur_memory [F,S,C];
        ref_pool [T',Q,C] / logit_pool [T',Q,classes]: "ref" fused queries and class logits of the
        frames that can serve as reference frames; others [F,R] (long): for each current frame the
        R rows of the pools that are ITS reference frames, in clip order; valid_ratio_cur [F,L,2].
        -> final_hs [F,Q,C], final_references [F,Q,4], aux outputs, top-k indices [F,k*R] (indices
        into the frame's own concatenation of R*Q reference queries, like the reference's)."""
        F_, Q, C = cur_hs.shape
        R = others.shape[1]
        # class-1 score of every reference query, per current frame: [F, R*Q]
        score = logit_pool.sigmoid()[:, :, 1][others].reshape(F_, R * Q)
        flat_pool = ref_pool.reshape(-1, C)
        ratios = valid_ratio_cur[:, :1].expand(F_, R, 2)       # R "levels" of reference boxes
        shapes1, lsi1 = spatial_shapes[0:1], level_start_index[0:1]
        for attr in ("_dfx_host", "_dfx_tokens"):
            if hasattr(spatial_shapes, attr) and spatial_shapes.shape[0] == 1:
                setattr(shapes1, attr, getattr(spatial_shapes, attr))
        aux, picks, pick_scores = [], [], []
        final_hs, final_refs = cur_hs, cur_reference_out
        # The scores do not change between the rounds and top-k comes back sorted: the picks of a round with a smaller k are
        # a prefix of the picks of the largest one (the reference calls topk on the same ref_prob_concat[:, :, 1] once per round, :530 / :557 / :582; equal scores may come
        # out in another order - as they may between any two topk implementations), so one selection serves all rounds.
        # the three temporal decoders cross-attend to the same current-frame memory: their value projections in one launch
        decoders = [getattr(self, f"temporal_decoder{i + 1}") for i in range(len(TOPK_PER_REF))]
        tdtd_values = None
        if cur_memory.is_cuda and not torch.is_grad_enabled() and all(len(d.layers) == 1 for d in decoders):
            from models.ops.modules.ms_deform_attn import project_values
            tdtd_values = project_values([d.layers[0].cross_attn for d in decoders], cur_memory, None)
        # ... and the rounds whose picks outnumber the pool project it instead of the picks (below): those projections of the one
        # pool in one launch as well
        tqes = [getattr(self, f"temporal_query_layer{i + 1}") for i in range(len(TOPK_PER_REF))]
        pooled = [F_ * k * R > flat_pool.shape[0] for k in TOPK_PER_REF]
        kv_pools = [None] * len(TOPK_PER_REF)
        if sum(pooled) >= 2 and cur_hs.is_cuda and not torch.is_grad_enabled():
            from . import fused_mha
            which = [i for i, on in enumerate(pooled) if on]
            if all(fused_mha.usable(tqes[i].cross_attn, flat_pool) for i in which) and tqes[0].cross_attn.embed_dim % 64 == 0:
                for i, kv in zip(which, fused_mha.project_kv_stack([tqes[i].cross_attn for i in which], flat_pool)):
                    kv_pools[i] = kv
        kmax = max(TOPK_PER_REF) * R
        vals_all, idx_all = torch.topk(score, kmax, dim=1)                          # [F,kmax R] in [0, R*Q)
        rows_all = torch.gather(others, 1, idx_all // Q) * Q + idx_all % Q          # rows of the flat pool
        for i, k in enumerate(TOPK_PER_REF):
            vals, idx, rows = vals_all[:, :k * R], idx_all[:, :k * R], rows_all[:, :k * R]
            picks.append(idx)
            pick_scores.append(vals)
            tqe = tqes[i]
            # the pool is shared by all current frames: its key / value projections are computed once per round and the picks
            # gather projected rows (F x k*R rows of two Linears become T*Q rows of one: 8x fewer at 32 frames), same values
            kv_pool = kv_pools[i] if kv_pools[i] is not None else (tqe.project_ref_pool(flat_pool) if pooled[i] else None)
            if kv_pool is not None:
                cur_hs = tqe(cur_hs, None, ref_kv=kv_pool[rows.reshape(-1)].view(F_, k * R, 2 * C))
            else:
                selected = flat_pool[rows.reshape(-1)].view(F_, k * R, C)
                cur_hs = tqe(cur_hs, selected)
            cur_hs, refs = decoders[i](cur_hs, cur_reference_out, cur_memory, shapes1, lsi1, ratios, None, None,
                                       values=None if tdtd_values is None else [tdtd_values[i]])
            if i < 2:
                aux.append({"pred_logits": temp_class_embed_list[i](cur_hs),
                            "pred_boxes": apply_box_head(temp_bbox_embed_list[i], cur_hs, refs)})
            else:
                final_hs, final_refs = cur_hs, refs
        self.last_pick_scores = pick_scores            # the scores of the picks (parity checks compare outside ties)
        return final_hs, final_refs, aux, picks

    # ------------------------------------------------------------------------------------------
    def forward(self, srcs, masks, pos_embeds, depth_srcs, depth_masks, depth_pos_embeds, imgs_whwh_shape,
                query_embed=None, class_embed=None, cur_bbox_embed=None, temp_class_embed_list=None,
                temp_bbox_embed_list=None, rgbd_query=[]):
        s = self._spatial_stage(srcs, masks, pos_embeds, depth_srcs, depth_masks, depth_pos_embeds, query_embed,
                                rgbd_query)
        hs, init_ref, inter_refs = s["hs"], s["init_reference"], s["inter_references"]
        if self.two_stage:
            return hs, init_ref, inter_refs, s["enc_outputs_class"], s["enc_outputs_coord_unact"]
        memory = s["memory"]
        if self.fixed_pretrained_model:
            memory, hs, inter_refs = memory.detach(), hs.detach(), inter_refs.detach()

        T = self.num_ref_frames + 1
        assert memory.shape[0] == T, f"expected a clip of {T} frames, got {memory.shape[0]}"
        whwh = torch.as_tensor(imgs_whwh_shape, dtype=torch.long, device=memory.device).repeat(1, self.num_query, 1)
        fs_cur = self.frame_stage(hs[-1][:1], inter_refs[-1][:1], memory[:1], s["lvl_pos_embed_flatten"][:1],
                                  s["last_hw"], whwh, class_embed, cur_bbox_embed, roles=("cur",))
        fs_ref = self.frame_stage(hs[-1][1:], inter_refs[-1][1:], memory[1:], s["lvl_pos_embed_flatten"][1:],
                                  s["last_hw"], whwh, class_embed, cur_bbox_embed, roles=("ref",))
        others = torch.arange(self.num_ref_frames, device=memory.device).view(1, -1)
        final_hs, final_refs, aux, _ = self.temporal_stage(
            fs_cur["cur"], inter_refs[-1][:1], memory[:1], fs_ref["ref"], fs_ref["logits"], others,
            s["spatial_shapes"], s["level_start_index"], s["valid_ratios"][:1], temp_class_embed_list,
            temp_bbox_embed_list)
        out = {"aux_outputs": aux}
        return hs[:, 0:1], init_ref[0:1], inter_refs[:, 0:1], None, None, final_hs, final_refs, out


def build_deforamble_transformer(args):
    return DeformableTransformer(
        d_model=args.hidden_dim, nhead=args.nheads, num_encoder_layers=args.enc_layers,
        num_decoder_layers=args.dec_layers, dim_feedforward=args.dim_feedforward, dropout=args.dropout,
        activation="relu", return_intermediate_dec=True, num_feature_levels=args.num_feature_levels,
        dec_n_points=args.dec_n_points, enc_n_points=args.enc_n_points, two_stage=args.two_stage,
        two_stage_num_proposals=args.num_queries, num_query=args.num_queries,
        n_temporal_decoder_layers=args.n_temporal_decoder_layers, num_ref_frames=args.num_ref_frames,
        fixed_pretrained_model=args.fixed_pretrained_model, args=args, use_depth=args.use_depth,
        depth_type=args.depth_type, dpth_n_points=args.dpth_n_points)
