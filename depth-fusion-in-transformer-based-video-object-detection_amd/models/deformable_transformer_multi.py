"""TransVOD transformer with RGB-D fusion (ref models/deformable_transformer_multi.py): the
spatial stage of the single-frame model on a clip (frame 0 current, frames 1..R reference)
followed by three temporal query encoder layers over the top-(k*R) reference queries and one
temporal decoder on the current frame's memory (ref forward :193-378).  ``TDAM`` is False in the
reference (:46), so ``temporal_encoder_layer`` holds weights but never runs.
"""
import torch

from .deformable_transformer_single import SpatialTransformerBase
from .transformer_layers import (DeformableTransformerDecoder, DeformableTransformerDecoderLayer,  # noqa: F401
                                 DeformableTransformerEncoder, DeformableTransformerEncoderLayer,
                                 DeformableTransformerFusionLayerV2, DepthDeformableTransformerEncoderLayer,
                                 RGBDDeformableTransformerEncoderV2, TemporalDeformableTransformerDecoder,
                                 TemporalDeformableTransformerEncoderLayer, TemporalQueryEncoder,
                                 TemporalQueryEncoderLayer, get_reference_points)

TOPK_PER_REF = (80, 50, 30)


class DeformableTransformer(SpatialTransformerBase):
    def __init__(self, d_model=256, nhead=8, num_encoder_layers=6, num_decoder_layers=6, dim_feedforward=1024,
                 dropout=0.1, activation="relu", return_intermediate_dec=False, num_feature_levels=4,
                 dec_n_points=4, enc_n_points=4, two_stage=False, two_stage_num_proposals=300,
                 n_temporal_decoder_layers=1, num_ref_frames=3, fixed_pretrained_model=False, args=None,
                 use_depth=False, depth_type="", dpth_feature_levels=1, dpth_n_points=4):
        super().__init__()
        self.num_ref_frames = num_ref_frames
        self.fixed_pretrained_model = fixed_pretrained_model
        self.n_temporal_query_layers = 3
        self.TDAM = False
        dec_layer = self._build_spatial(d_model, nhead, num_encoder_layers, num_decoder_layers, dim_feedforward,
                                        dropout, activation, return_intermediate_dec, num_feature_levels,
                                        dec_n_points, enc_n_points, two_stage, two_stage_num_proposals, use_depth,
                                        depth_type, dpth_feature_levels, dpth_n_points)
        self.temporal_encoder_layer = TemporalDeformableTransformerEncoderLayer(
            d_model, dim_feedforward, dropout, activation, num_ref_frames, nhead, enc_n_points)
        for i in (1, 2, 3):
            setattr(self, f"temporal_query_layer{i}",
                    TemporalQueryEncoderLayer(d_model, dim_feedforward, dropout, activation, nhead))
        self.temporal_decoder = TemporalDeformableTransformerDecoder(dec_layer, n_temporal_decoder_layers, False)
        self._reset_parameters()

    def forward(self, srcs, masks, pos_embeds, depth_srcs, depth_masks, depth_pos_embeds, query_embed=None,
                class_embed=None, rgbd_query=[]):
        s = self._spatial_stage(srcs, masks, pos_embeds, depth_srcs, depth_masks, depth_pos_embeds, query_embed,
                                rgbd_query)
        hs, init_ref, inter_refs = s["hs"], s["init_reference"], s["inter_references"]
        if self.two_stage:
            return hs, init_ref, inter_refs, s["enc_outputs_class"], s["enc_outputs_coord_unact"]
        memory = s["memory"]
        if self.fixed_pretrained_model:
            memory, hs, inter_refs = memory.detach(), hs.detach(), inter_refs.detach()
        R = self.num_ref_frames
        assert memory.shape[0] == R + 1
        cur_memory = memory[:1]
        shapes, lsi = s["spatial_shapes"], s["level_start_index"]
        ratios = s["valid_ratios"][0:1].expand(1, R, 2)
        if self.TDAM:   # never taken in the reference configuration; kept for completeness
            ref_memory = torch.cat(list(memory[1:].unsqueeze(1)), 1) + \
                torch.cat(list(s["lvl_pos_embed_flatten"][1:].unsqueeze(1)), 1)
            ref_shapes = shapes.expand(R, 2).contiguous()
            frame_start = torch.cat((ref_shapes.new_zeros((1,)), ref_shapes.prod(1).cumsum(0)[:-1])).contiguous()
            grid = get_reference_points(shapes, ratios, device=cur_memory.device)
            cur_memory = self.temporal_encoder_layer(cur_memory, s["lvl_pos_embed_flatten"][0:1], grid, ref_memory,
                                                     ref_shapes, frame_start)
        Q, C = hs.shape[2], hs.shape[3]
        cur_hs = hs[-1][:1]
        ref_hs = hs[-1][1:].reshape(1, R * Q, C)
        cur_reference_out = inter_refs[-1][:1]
        logits = class_embed(ref_hs)
        ncls = logits.shape[2] - 1
        score = logits.sigmoid()[:, :, :-1].reshape(1, -1)     # every class but the last, flattened
        for i, k in enumerate(TOPK_PER_REF):
            idx = torch.topk(score, k * R, dim=1)[1] // ncls
            selected = torch.gather(ref_hs, 1, idx.unsqueeze(-1).repeat(1, 1, C))
            cur_hs = getattr(self, f"temporal_query_layer{i + 1}")(cur_hs, selected)
        shapes1, lsi1 = shapes[0:1], lsi[0:1]
        for attr in ("_dfx_host", "_dfx_tokens"):
            if hasattr(shapes, attr) and shapes.shape[0] == 1:
                setattr(shapes1, attr, getattr(shapes, attr))
        final_hs, final_refs = self.temporal_decoder(cur_hs, cur_reference_out, cur_memory, shapes1, lsi1, ratios,
                                                     None, None)
        return hs[:, 0:1], init_ref[0:1], inter_refs[:, 0:1], None, None, final_hs, final_refs


def build_deforamble_transformer(args):
    return DeformableTransformer(
        d_model=args.hidden_dim, nhead=args.nheads, num_encoder_layers=args.enc_layers,
        num_decoder_layers=args.dec_layers, dim_feedforward=args.dim_feedforward, dropout=args.dropout,
        activation="relu", return_intermediate_dec=True, num_feature_levels=args.num_feature_levels,
        dec_n_points=args.dec_n_points, enc_n_points=args.enc_n_points, two_stage=args.two_stage,
        two_stage_num_proposals=args.num_queries, n_temporal_decoder_layers=args.n_temporal_decoder_layers,
        num_ref_frames=args.num_ref_frames, fixed_pretrained_model=args.fixed_pretrained_model, args=args,
        use_depth=args.use_depth, depth_type=args.depth_type, dpth_n_points=args.dpth_n_points)
