"""Single-frame deformable transformer with RGB-D fusion
(ref models/deformable_transformer_single.py).  ``SpatialTransformerBase`` holds the stage that
is identical in all three reference transformers - level flattening, Late Fusion / Encoder Cross
Fusion dispatch, encoder, query split, decoder - and is reused by the TransVOD / TransVOD++
files; ``DeformableTransformer`` here is the single-frame model (forward ref :179-338).
"""
import math

import torch
from torch import nn

from util.memo import memo_on

from models.fused import Linear
from torch.nn.init import constant_, normal_, xavier_uniform_

from models.ops.modules import MSDeformAttn

from .transformer_layers import (DeformableTransformerDecoder, DeformableTransformerDecoderLayer,  # noqa: F401
                                 DeformableTransformerEncoder, DeformableTransformerEncoderLayer,
                                 DeformableTransformerFusionLayerV2, DepthDeformableTransformerEncoderLayer,
                                 RGBDDeformableTransformerEncoderV2, _get_activation_fn, _get_clones,
                                 get_reference_points, get_valid_ratio, make_level_tensors)


class SpatialTransformerBase(nn.Module):
    """Parameters: ``encoder``, ``decoder``, ``level_embed``, ``reference_points`` (or the two-stage
    heads) and, by fusion type, ``depth_encoder_layer`` (Late Fusion) / ``encoder.fusion_layers``
    (Encoder Cross Fusion) - the reference's names."""

    def _build_spatial(self, d_model, nhead, num_encoder_layers, num_decoder_layers, dim_feedforward, dropout,
                       activation, return_intermediate_dec, num_feature_levels, dec_n_points, enc_n_points,
                       two_stage, two_stage_num_proposals, use_depth, depth_type, dpth_feature_levels,
                       dpth_n_points):
        self.use_depth, self.depth_type = use_depth, depth_type
        self.residual_fusion = "noresidual" not in depth_type
        self.rgbd_query = "concat" in depth_type
        self.d_model, self.nhead = d_model, nhead
        self.two_stage, self.two_stage_num_proposals = two_stage, two_stage_num_proposals
        self.depth_self_attn, self.late_fusion_layers = True, 1
        self.adaptation_layers, self.gate, self.encoder_cross_fusion = True, True, True

        enc_layer = DeformableTransformerEncoderLayer(d_model, dim_feedforward, dropout, activation,
                                                      num_feature_levels, nhead, enc_n_points)
        if "encoder_cf" in depth_type:
            self.num_depth_encoder_layers, self.num_enc_fusion_layers = 4, 4
            self.enc_fusion_layers_order = [0, 1, 2, 3]
            fusion_layer = DeformableTransformerFusionLayerV2(d_model, dim_feedforward, dropout, activation,
                                                              num_feature_levels, nhead, enc_n_points)
            self.encoder = RGBDDeformableTransformerEncoderV2(enc_layer, fusion_layer, num_encoder_layers,
                                                              self.num_depth_encoder_layers,
                                                              self.num_enc_fusion_layers, self.enc_fusion_layers_order)
        else:
            self.encoder = DeformableTransformerEncoder(enc_layer, num_encoder_layers)
        if "latefusion" in depth_type:
            self.depth_encoder_layer = DepthDeformableTransformerEncoderLayer(
                d_model, dim_feedforward, dropout, activation, dpth_feature_levels, nhead, dpth_n_points,
                self.depth_self_attn, self.gate, self.adaptation_layers)
        dec_layer = DeformableTransformerDecoderLayer(d_model, dim_feedforward, dropout, activation,
                                                      num_feature_levels, nhead, dec_n_points)
        self.decoder = DeformableTransformerDecoder(dec_layer, num_decoder_layers, return_intermediate_dec)
        self.level_embed = nn.Parameter(torch.Tensor(num_feature_levels, d_model))
        if two_stage:
            self.enc_output = Linear(d_model, d_model)
            self.enc_output_norm = nn.LayerNorm(d_model)
            self.pos_trans = Linear(d_model * 2, d_model * 2)
            self.pos_trans_norm = nn.LayerNorm(d_model * 2)
        else:
            self.reference_points = Linear(d_model, 2)
        return dec_layer

    def _reset_parameters(self):
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        for m in self.modules():
            if isinstance(m, MSDeformAttn):
                m._reset_parameters()
            elif isinstance(m, nn.LayerNorm):
                constant_(m.weight, 1.0)
                constant_(m.bias, 0.0)
        if not self.two_stage:
            xavier_uniform_(self.reference_points.weight.data, gain=1.0)
            constant_(self.reference_points.bias.data, 0.)
        normal_(self.level_embed)

    get_valid_ratio = staticmethod(get_valid_ratio)
    get_reference_points = staticmethod(get_reference_points)

    # ---- two-stage helpers (ref :125-163) ----
    def get_proposal_pos_embed(self, proposals):
        num_pos_feats, temperature, scale = 128, 10000, 2 * math.pi
        k = torch.arange(num_pos_feats, dtype=torch.float32, device=proposals.device)
        freq = temperature ** (2 * (k // 2) / num_pos_feats)
        pos = (proposals.sigmoid() * scale)[:, :, :, None] / freq
        return torch.stack((pos[:, :, :, 0::2].sin(), pos[:, :, :, 1::2].cos()), dim=4).flatten(2)

    def gen_encoder_output_proposals(self, memory, memory_padding_mask, spatial_shapes):
        from .transformer_layers import host_shapes
        N = memory.shape[0]
        proposals, cur = [], 0
        for lvl, (H, W) in enumerate(host_shapes(spatial_shapes)):
            m = memory_padding_mask[:, cur:cur + H * W].view(N, H, W, 1)
            valid_h = torch.sum(~m[:, :, 0, 0], 1)
            valid_w = torch.sum(~m[:, 0, :, 0], 1)
            gy, gx = torch.meshgrid(torch.linspace(0, H - 1, H, dtype=torch.float32, device=memory.device),
                                    torch.linspace(0, W - 1, W, dtype=torch.float32, device=memory.device),
                                    indexing="ij")
            grid = torch.cat([gx.unsqueeze(-1), gy.unsqueeze(-1)], -1)
            scale = torch.cat([valid_w.unsqueeze(-1), valid_h.unsqueeze(-1)], 1).view(N, 1, 1, 2)
            grid = (grid.unsqueeze(0).expand(N, -1, -1, -1) + 0.5) / scale
            wh = torch.ones_like(grid) * 0.05 * (2.0 ** lvl)
            proposals.append(torch.cat((grid, wh), -1).view(N, -1, 4))
            cur += H * W
        props = torch.cat(proposals, 1)
        valid = ((props > 0.01) & (props < 0.99)).all(-1, keepdim=True)
        props = torch.log(props / (1 - props))
        props = props.masked_fill(memory_padding_mask.unsqueeze(-1), float("inf")).masked_fill(~valid, float("inf"))
        mem = memory.masked_fill(memory_padding_mask.unsqueeze(-1), float(0)).masked_fill(~valid, float(0))
        return self.enc_output_norm(self.enc_output(mem)), props

    # ---- the shared spatial stage ----
    @staticmethod
    def _flatten_levels(feats, masks, pos_embeds, level_embed=None):
        tok, msk, pos, shapes = [], [], [], []
        for lvl, (f, m, p) in enumerate(zip(feats, masks, pos_embeds)):
            shapes.append((f.shape[2], f.shape[3]))
            tok.append(f.flatten(2).transpose(1, 2))
            msk.append(m.flatten(1))
            if level_embed is None:
                pos.append(p.flatten(2).transpose(1, 2))
            else:
                # positional embedding + level embedding: derived from the (memoised) positional tensor and the
                # parameter's version, so inference builds it once per mask (util/memo.py)
                pos.append(memo_on(p, ("lvl_pos", lvl, level_embed.data_ptr(), level_embed._version),
                                   lambda p=p, lvl=lvl: p.flatten(2).transpose(1, 2) + level_embed[lvl].view(1, 1, -1)))
        one = len(tok) == 1                     # a single level: nothing to concatenate (torch.cat would copy)
        return (tok[0] if one else torch.cat(tok, 1), msk[0] if one else torch.cat(msk, 1),
                pos[0] if one else torch.cat(pos, 1), shapes)

    def _spatial_stage(self, srcs, masks, pos_embeds, depth_srcs, depth_masks, depth_pos_embeds, query_embed,
                       rgbd_query=()):
        """-> dict(hs, init_reference, inter_references, memory, mask_flatten, lvl_pos_embed_flatten,
        spatial_shapes, level_start_index, valid_ratios, last_hw, enc_outputs_*)."""
        assert self.two_stage or query_embed is not None
        src, mask, lvl_pos, shape_list = self._flatten_levels(srcs, masks, pos_embeds, self.level_embed)
        rgbd = torch.cat([q.flatten(2).transpose(1, 2) for q in rgbd_query], 1) if len(rgbd_query) > 0 else None
        last_hw = shape_list[-1]
        spatial_shapes, level_start_index = make_level_tensors(shape_list, src.device)
        valid_ratios = torch.stack([get_valid_ratio(m) for m in masks], 1)

        depth = None
        wants_depth = self.use_depth and ("latefusion" in self.depth_type or "encoder_cf" in self.depth_type)
        if wants_depth:
            assert depth_srcs is not None and depth_masks is not None and depth_pos_embeds is not None, \
                "Depth information is required for Deformable DETR with depth"
            assert len(depth_srcs) == len(depth_masks) == len(depth_pos_embeds), \
                "The number of depth sources, masks and pos_embeds should be the same"
            d_src, d_mask, d_pos, d_shape_list = self._flatten_levels(depth_srcs, depth_masks, depth_pos_embeds)
            last_hw = d_shape_list[-1]     # the reference's h, w end up being those of the last depth level
            d_shapes, d_lsi = make_level_tensors(d_shape_list, d_src.device)
            d_ratios = torch.stack([get_valid_ratio(m) for m in depth_masks], 1)
            depth = (d_src, d_mask, d_pos, d_shapes, d_lsi, d_ratios)

        if depth is not None and "latefusion" in self.depth_type:
            d_src, d_mask, d_pos, d_shapes, d_lsi, d_ratios = depth
            rgb_ref = get_reference_points(spatial_shapes, valid_ratios, device=src.device)
            depth_ref = None   # computed by the reference, never used by the layer
            fused = self.depth_encoder_layer(src, lvl_pos, d_pos, spatial_shapes, rgb_ref, depth_ref, d_src,
                                             d_shapes, d_lsi, mask, d_mask)
            src = src + fused

        if depth is not None and "encoder_cf" in self.depth_type:
            d_src, d_mask, d_pos, d_shapes, d_lsi, d_ratios = depth
            memory = self.encoder(src, spatial_shapes, level_start_index, valid_ratios, lvl_pos, mask,
                                  rgbd if self.rgbd_query else None, d_src, d_shapes, d_lsi, d_ratios, d_pos, d_mask)
        elif isinstance(self.encoder, RGBDDeformableTransformerEncoderV2):
            memory = self.encoder(src, spatial_shapes, level_start_index, valid_ratios, lvl_pos, mask)
        else:
            memory = self.encoder(src, spatial_shapes, level_start_index, valid_ratios, lvl_pos, mask,
                                  rgbd if self.rgbd_query else None)

        bs, _, c = memory.shape
        enc_cls = enc_coord = None
        if self.two_stage:
            out_mem, out_props = self.gen_encoder_output_proposals(memory, mask, spatial_shapes)
            enc_cls = self.decoder.class_embed[self.decoder.num_layers](out_mem)
            enc_coord = self.decoder.bbox_embed[self.decoder.num_layers](out_mem) + out_props
            top = torch.topk(enc_cls[..., 0], self.two_stage_num_proposals, dim=1)[1]
            coords = torch.gather(enc_coord, 1, top.unsqueeze(-1).repeat(1, 1, 4)).detach()
            reference_points = coords.sigmoid()
            pos_trans = self.pos_trans_norm(self.pos_trans(self.get_proposal_pos_embed(coords)))
            query_pos, tgt = torch.split(pos_trans, c, dim=2)
        else:
            query_pos, tgt = torch.split(query_embed, c, dim=1)
            query_pos = query_pos.unsqueeze(0).expand(bs, -1, -1)
            tgt = tgt.unsqueeze(0).expand(bs, -1, -1)
            reference_points = self.reference_points(query_pos).sigmoid()
        hs, inter_refs = self.decoder(tgt, reference_points, memory, spatial_shapes, level_start_index,
                                      valid_ratios, query_pos, mask)
        return dict(hs=hs, init_reference=reference_points, inter_references=inter_refs, memory=memory,
                    mask_flatten=mask, lvl_pos_embed_flatten=lvl_pos, spatial_shapes=spatial_shapes,
                    level_start_index=level_start_index, valid_ratios=valid_ratios, last_hw=last_hw,
                    enc_outputs_class=enc_cls, enc_outputs_coord_unact=enc_coord)


class DeformableTransformer(SpatialTransformerBase):
    def __init__(self, d_model=256, nhead=8, num_encoder_layers=6, num_decoder_layers=6, dim_feedforward=1024,
                 dropout=0.1, activation="relu", return_intermediate_dec=False, num_feature_levels=4,
                 dec_n_points=4, enc_n_points=4, two_stage=False, two_stage_num_proposals=300, use_depth=False,
                 depth_type="Baseline_rgb", dpth_feature_levels=1, dpth_n_points=4):
        super().__init__()
        self._build_spatial(d_model, nhead, num_encoder_layers, num_decoder_layers, dim_feedforward, dropout,
                            activation, return_intermediate_dec, num_feature_levels, dec_n_points, enc_n_points,
                            two_stage, two_stage_num_proposals, use_depth, depth_type, dpth_feature_levels,
                            dpth_n_points)
        self._reset_parameters()

    def forward(self, srcs, masks, pos_embeds, depth_srcs, depth_masks, depth_pos_embeds, query_embed=None,
                rgbd_query=[]):
        s = self._spatial_stage(srcs, masks, pos_embeds, depth_srcs, depth_masks, depth_pos_embeds, query_embed,
                                rgbd_query)
        return (s["hs"], s["init_reference"], s["inter_references"], s["enc_outputs_class"],
                s["enc_outputs_coord_unact"])


def build_deforamble_transformer(args):
    return DeformableTransformer(
        d_model=args.hidden_dim, nhead=args.nheads, num_encoder_layers=args.enc_layers,
        num_decoder_layers=args.dec_layers, dim_feedforward=args.dim_feedforward, dropout=args.dropout,
        activation="relu", return_intermediate_dec=True, num_feature_levels=args.num_feature_levels,
        dec_n_points=args.dec_n_points, enc_n_points=args.enc_n_points, two_stage=args.two_stage,
        two_stage_num_proposals=args.num_queries, use_depth=args.use_depth, depth_type=args.depth_type,
        dpth_n_points=args.dpth_n_points)
