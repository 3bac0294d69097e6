"""Shared case recipes for the module / transformer level golden vectors.

``run_cases(ns)`` builds every block on the hot path from the module namespace ``ns``, fills
its parameters by state_dict name (tests/_param_fill.py), feeds seeded inputs and returns the
outputs.  tools/gen_golden_models.py calls it with the REFERENCE's modules (build container
only) and stores the outputs in tests/golden/models.npz; tests/test_models_golden.py calls it
with this repository's modules and compares.  Because one recipe drives both sides, the
fixtures pin constructor signatures, call signatures and state_dict keys as well as numerics.

``ns`` attributes: MSDeformAttn, ts / tpp / tm (the three transformer modules), RCNNHead, dfb
(dformer_backbone), dcf (dformer_crossfusion_backbone), PositionEmbeddingSine, NestedTensor,
inverse_sigmoid.
"""
from types import SimpleNamespace

import torch

from tests._param_fill import fill_params_by_name as _fill_by_name


_DEVICE = "cpu"   # set by run_cases; inputs are always DRAWN on the CPU generator, then moved


def rnd(seed, *shape, scale=1.0):
    return (torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale).to(_DEVICE)


def urnd(seed, *shape):
    return torch.rand(*shape, generator=torch.Generator().manual_seed(seed)).to(_DEVICE)


def levels(shape_list):
    shapes = torch.as_tensor(shape_list, dtype=torch.long).to(_DEVICE)
    lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
    return shapes, lsi, int(shapes.prod(1).sum())


def pad_masks(seed, n, h, w):
    """Image-like padding masks: frame i is valid on [0:vh, 0:vw]."""
    g = torch.Generator().manual_seed(seed)
    m = torch.ones(n, h, w, dtype=torch.bool)
    for i in range(n):
        vh = int(torch.randint(max(1, h * 2 // 3), h + 1, (1,), generator=g))
        vw = int(torch.randint(max(1, w * 2 // 3), w + 1, (1,), generator=g))
        m[i, :vh, :vw] = False
    m[0] = False
    return m.to(_DEVICE)



def run_cases(ns, device="cpu"):
    MSDeformAttn, ts, tpp, tm, RCNNHead = ns.MSDeformAttn, ns.ts, ns.tpp, ns.tm, ns.RCNNHead
    dfb, dcf, PositionEmbeddingSine = ns.dfb, ns.dcf, ns.PositionEmbeddingSine
    NestedTensor, inverse_sigmoid = ns.NestedTensor, ns.inverse_sigmoid
    global _DEVICE
    _DEVICE = device
    blobs = {}

    def fill_params_by_name(module, seed=0, prefix=""):   # fill on the CPU, then move
        return _fill_by_name(module, seed=seed, prefix=prefix).to(device)

    def put(case, **tensors):
        for k, v in tensors.items():
            blobs[f"{case}.{k}"] = v.detach().cpu()

    # ---- a4: MSDeformAttn module, 2-d and 4-d reference points, padding mask ---------------------------
    for case, L, ref_dim, shp in (("attn_2d", 2, 2, [(6, 7), (3, 4)]), ("attn_4d", 1, 4, [(5, 8)])):
        shapes, lsi, S = levels(shp)
        mod = fill_params_by_name(MSDeformAttn(256, L, 8, 4).eval(), seed=1)
        q, x = rnd(1, 2, 37, 256), rnd(2, 2, S, 256)
        ref = urnd(3, 2, 37, L, ref_dim)
        if ref_dim == 4:
            ref[..., 2:] *= 0.4
        mask = urnd(4, 2, S) > 0.85
        put(case, out=mod(q, ref, x, shapes, lsi, mask))

    # ---- a5/a9/TQE/a12: single layers -------------------------------------------------------------------
    shapes, lsi, S = levels([(6, 8)])
    grid = ts.DeformableTransformer.get_reference_points(shapes, torch.ones(2, 1, 2, device=device), device)
    src, pos = rnd(10, 2, S, 256), rnd(11, 2, S, 256)
    layer = fill_params_by_name(ts.DeformableTransformerEncoderLayer(256, 1024, 0.1, "relu", 1, 8, 4).eval(), seed=2)
    put("enc_layer", out=layer(src, pos, grid, shapes, lsi, None))

    tgt, qpos = rnd(12, 2, 21, 256), rnd(13, 2, 21, 256)
    ref4 = urnd(14, 2, 21, 1, 4) * torch.tensor([1, 1, 0.4, 0.4], device=device)
    layer = fill_params_by_name(ts.DeformableTransformerDecoderLayer(256, 1024, 0.1, "relu", 1, 8, 4).eval(), seed=3)
    put("dec_layer", out=layer(tgt, qpos, ref4, src, shapes, lsi, None))

    depth = rnd(15, 2, S, 256)
    layer = fill_params_by_name(ts.DepthDeformableTransformerEncoderLayer(256, 1024, 0.1, "relu", 1, 8, 4, True, True, True).eval(), seed=4)
    put("latefusion_layer", out=layer(src, pos, None, shapes, grid, None, depth, shapes, lsi, None, None))

    layer = fill_params_by_name(ts.DeformableTransformerFusionLayerV2(256, 1024, 0.1, "relu", 1, 8, 4).eval(), seed=5)
    put("fusion_v2_layer", out=layer(src, pos, grid, depth, shapes, lsi, None))

    refq = rnd(16, 1, 55, 256)
    layer = fill_params_by_name(tpp.TemporalQueryEncoderLayer(256, 1024, 0.1, "relu", 8).eval(), seed=6)
    put("tqe_layer", out=layer(tgt[:1], refq))

    cfg = {"MODEL": {"SparseRCNN": {"NHEADS": 8, "DROPOUT": 0.0, "DIM_FEEDFORWARD": 2048, "ACTIVATION": "relu",
                                    "HIDDEN_DIM": 256, "NUM_CLS": 1, "NUM_REG": 3, "NUM_HEADS": 6, "NUM_DYNAMIC": 2,
                                    "DIM_DYNAMIC": 64}, "ROI_BOX_HEAD": {"POOLER_RESOLUTION": 7}}}
    head = fill_params_by_name(RCNNHead(cfg, 256, 3, 1024, 8, 0.1, "relu").eval(), seed=7)
    roi_feats, props = rnd(17, 21, 256, 7, 7), rnd(18, 1, 21, 256)
    put("rcnn_head", out=head(roi_feats, props))

    # ---- a8: backbone cross-fusion block ---------------------------------------------------------------
    layer = fill_params_by_name(dcf.DepthDeformableTransformerEncoderLayer(256, 1024, 0.1, "relu", 1, 8, 4).eval(), seed=8)
    rgb_map, d_map = rnd(19, 1, 256, 5, 7), rnd(20, 1, 256, 9, 13)
    pe = PositionEmbeddingSine(128, normalize=True)
    m_rgb, m_d = pad_masks(21, 1, 5, 7), pad_masks(22, 1, 9, 13)
    m_rgb[0, :, 6:] = True
    m_d[0, :, 11:] = True
    p_rgb, p_d = pe(NestedTensor(rgb_map, m_rgb)), pe(NestedTensor(d_map, m_d))
    put("fuse_layers", out=dcf.FusionBackboneBase.fuse_layers(rgb_map, d_map, p_rgb, p_d, m_rgb, m_d, layer))

    # ---- a17: positional encoding, inverse_sigmoid ------------------------------------------------------
    m = pad_masks(23, 2, 7, 9)
    put("pos_sine", out=pe(NestedTensor(torch.zeros(2, 1, 7, 9, device=device), m)))
    x = torch.cat([urnd(24, 50), torch.tensor([0.0, 1.0, -0.2, 1.3, 1e-7, 1 - 1e-7], device=device)])
    put("inverse_sigmoid", out=inverse_sigmoid(x))

    # ---- a15: DFormer depth backbone --------------------------------------------------------------------
    args = SimpleNamespace(hidden_dim=256, position_embedding="sine", dformer_weights=None)
    dback = fill_params_by_name(dfb.build_dformer_backbone(args).eval(), seed=9)
    dimg, dmask = rnd(25, 2, 1, 64, 96), pad_masks(26, 2, 64, 96)
    feats, dpos = dback(NestedTensor(dimg, dmask))
    put("dformer", feat=feats[0].tensors, feat_mask=feats[0].mask, pos=dpos[0])


    # ---- a10: full single-frame transformers -----------------------------------------------------------
    def mlp(seed):
        import torch.nn as nn

        class MLP(nn.Module):
            def __init__(self):
                super().__init__()
                self.num_layers = 3
                self.layers = nn.ModuleList([nn.Linear(256, 256), nn.Linear(256, 256), nn.Linear(256, 4)])

            def forward(self, x):
                for i, l in enumerate(self.layers):
                    x = torch.relu(l(x)) if i < 2 else l(x)
                return x
        return MLP()


    def box_heads(n, seed):
        heads = torch.nn.ModuleList([mlp(0) for _ in range(n)])
        fill_params_by_name(heads, seed=seed, prefix="bbox_embed.")
        for h in heads:                       # keep refined boxes well inside (0,1)
            h.layers[-1].weight.mul_(0.2)
        return heads.eval()


    def transformer_inputs(seed, T, shp, with_depth, d_shp=None):
        srcs = [rnd(seed + i, T, 256, h, w) for i, (h, w) in enumerate(shp)]
        masks = [pad_masks(seed + 10 + i, T, h, w) for i, (h, w) in enumerate(shp)]
        poss = [pe(NestedTensor(s, m)) for s, m in zip(srcs, masks)]
        if not with_depth:
            return srcs, masks, poss, [], [], []
        d_shp = d_shp or shp[:1]
        dsrcs = [rnd(seed + 20 + i, T, 256, h, w) for i, (h, w) in enumerate(d_shp)]
        dmasks = [masks[0] if tuple(d_shp[0]) == tuple(shp[0]) else pad_masks(seed + 30, T, *d_shp[0])]
        dposs = [pe(NestedTensor(s, m)) for s, m in zip(dsrcs, dmasks)]
        return srcs, masks, poss, dsrcs, dmasks, dposs


    def flat(prefix, tensors):
        return {f"{prefix}{i}": t for i, t in enumerate(tensors)}


    for case, dtype_str, L, shp in (("single_baseline", "Baseline_rgb", 1, [(6, 8)]),
                                    ("single_latefusion", "DepthDeform_latefusion_dformer", 1, [(6, 8)]),
                                    ("single_encodercf", "DepthDeform_encoder_cf_dformer", 1, [(6, 8)]),
                                    ("single_baseline_l4", "Baseline_rgb", 4, [(8, 10), (4, 5), (2, 3), (1, 2)])):
        use_depth = dtype_str != "Baseline_rgb"
        n_enc = 4 if "encoder_cf" in dtype_str else 2
        tr = ts.DeformableTransformer(d_model=256, nhead=8, num_encoder_layers=n_enc, num_decoder_layers=2,
                                      dim_feedforward=1024, dropout=0.1, activation="relu", return_intermediate_dec=True,
                                      num_feature_levels=L, dec_n_points=4, enc_n_points=4, two_stage=False,
                                      two_stage_num_proposals=30, use_depth=use_depth, depth_type=dtype_str,
                                      dpth_n_points=4).eval()
        fill_params_by_name(tr, seed=11)
        tr.decoder.bbox_embed = box_heads(2, seed=12)
        srcs, masks, poss, dsrcs, dmasks, dposs = transformer_inputs(40, 2, shp, use_depth)
        qe = rnd(60, 30, 512)
        hs, init_ref, inter, _, _ = tr(srcs, masks, poss, dsrcs, dmasks, dposs, qe, [])
        put(case, hs=hs, init_ref=init_ref, inter_refs=inter)

    # ---- a11: TransVOD++ transformer (RoIAlign = the oracle restatement on both sides) ------------------
    R, Q = 2, 90
    for case, dtype_str in (("multipp_latefusion", "DepthDeform_latefusion_dformer"), ("multipp_baseline", "Baseline_rgb")):
        use_depth = dtype_str != "Baseline_rgb"
        tr = tpp.DeformableTransformer(d_model=256, nhead=8, num_encoder_layers=2, num_decoder_layers=2,
                                       dim_feedforward=1024, dropout=0.1, activation="relu", return_intermediate_dec=True,
                                       num_feature_levels=1, dec_n_points=4, enc_n_points=4, two_stage=False,
                                       two_stage_num_proposals=Q, num_query=Q, n_temporal_decoder_layers=1,
                                       num_ref_frames=R, fixed_pretrained_model=False, args=None, use_depth=use_depth,
                                       depth_type=dtype_str, dpth_n_points=4).eval()
        fill_params_by_name(tr, seed=21)
        tr.decoder.bbox_embed = box_heads(2, seed=22)
        class_embed = fill_params_by_name(torch.nn.Linear(256, 3), seed=23, prefix="class_embed.")
        tcls = torch.nn.ModuleList([torch.nn.Linear(256, 3) for _ in range(3)])
        fill_params_by_name(tcls, seed=24, prefix="temp_class_embed_list.")
        tbox = torch.nn.ModuleList([mlp(0) for _ in range(3)])
        fill_params_by_name(tbox, seed=25, prefix="temp_bbox_embed_list.")
        srcs, masks, poss, dsrcs, dmasks, dposs = transformer_inputs(70, R + 1, [(6, 8)], use_depth)
        masks = [torch.zeros_like(m) for m in masks]          # clips are same-size frames: no padding
        dmasks = [torch.zeros_like(m) for m in dmasks]
        poss = [pe(NestedTensor(s, m)) for s, m in zip(srcs, masks)]
        dposs = [pe(NestedTensor(s, m)) for s, m in zip(dsrcs, dmasks)]
        qe = rnd(90, Q, 512)
        whwh = (128, 96, 128, 96)       # image of 96 x 128 pixels, stride-16 map 6 x 8
        res = tr(srcs, masks, poss, dsrcs, dmasks, dposs, whwh, qe, class_embed, tr.decoder.bbox_embed[-1], tcls, tbox, [])
        hs0, init0, inter0, _, _, final_hs, final_refs, out = res
        put(case, hs=hs0, init_ref=init0, inter_refs=inter0, final_hs=final_hs, final_refs=final_refs,
            aux0_logits=out["aux_outputs"][0]["pred_logits"], aux0_boxes=out["aux_outputs"][0]["pred_boxes"],
            aux1_logits=out["aux_outputs"][1]["pred_logits"], aux1_boxes=out["aux_outputs"][1]["pred_boxes"])

    # ---- TransVOD transformer ----------------------------------------------------------------------------
    tr = tm.DeformableTransformer(d_model=256, nhead=8, num_encoder_layers=2, num_decoder_layers=2, dim_feedforward=1024,
                                  dropout=0.1, activation="relu", return_intermediate_dec=True, num_feature_levels=1,
                                  dec_n_points=4, enc_n_points=4, two_stage=False, two_stage_num_proposals=Q,
                                  n_temporal_decoder_layers=1, num_ref_frames=R, fixed_pretrained_model=False, args=None,
                                  use_depth=False, depth_type="Baseline_rgb", dpth_n_points=4).eval()
    fill_params_by_name(tr, seed=31)
    tr.decoder.bbox_embed = box_heads(2, seed=32)
    class_embed = fill_params_by_name(torch.nn.Linear(256, 3), seed=33, prefix="class_embed.")
    srcs, masks, poss, _, _, _ = transformer_inputs(110, R + 1, [(6, 8)], False)
    masks = [torch.zeros_like(m) for m in masks]
    poss = [pe(NestedTensor(s, m)) for s, m in zip(srcs, masks)]
    qe = rnd(120, Q, 512)
    res = tr(srcs, masks, poss, [], [], [], qe, class_embed, [])
    put("multi_baseline", hs=res[0], final_hs=res[5], final_refs=res[6])

    # ---- TransVOD transformer + Late Fusion (deformable_transformer_multi.py:193-378 with use_depth) -----------------
    tr = tm.DeformableTransformer(d_model=256, nhead=8, num_encoder_layers=2, num_decoder_layers=2, dim_feedforward=1024,
                                  dropout=0.1, activation="relu", return_intermediate_dec=True, num_feature_levels=1,
                                  dec_n_points=4, enc_n_points=4, two_stage=False, two_stage_num_proposals=Q,
                                  n_temporal_decoder_layers=1, num_ref_frames=R, fixed_pretrained_model=False, args=None,
                                  use_depth=True, depth_type="DepthDeform_latefusion_dformer", dpth_n_points=4).eval()
    fill_params_by_name(tr, seed=35)
    tr.decoder.bbox_embed = box_heads(2, seed=36)
    class_embed = fill_params_by_name(torch.nn.Linear(256, 3), seed=37, prefix="class_embed.")
    srcs, masks, poss, dsrcs, dmasks, dposs = transformer_inputs(130, R + 1, [(6, 8)], True)
    masks = [torch.zeros_like(m) for m in masks]
    dmasks = [torch.zeros_like(m) for m in dmasks]
    poss = [pe(NestedTensor(s, m)) for s, m in zip(srcs, masks)]
    dposs = [pe(NestedTensor(s, m)) for s, m in zip(dsrcs, dmasks)]
    qe = rnd(140, Q, 512)
    res = tr(srcs, masks, poss, dsrcs, dmasks, dposs, qe, class_embed, [])
    put("multi_latefusion", hs=res[0], init_ref=res[1], inter_refs=res[2], final_hs=res[5], final_refs=res[6])

    return blobs


def run_train_cases(ns):
    """The blocks' TRAIN-mode forward (sub-layer Dropouts active, p = 0.2 as every shipped config sets it) under a fixed
    global seed, CPU only: the reference applies ``norm(residual + dropout(sublayer(x)))``
    (deformable_transformer_single.py:375,396,545-557,625-643; deformable_transformer_multi_plusplus.py:815-838;
    sparse_roi_head/head.py:75-80).  Both sides draw their masks from the same generator state, so an implementation that
    skips a Dropout, applies it to another tensor layout or in another order differs at once."""
    ts, tpp, RCNNHead = ns.ts, ns.tpp, ns.RCNNHead
    global _DEVICE
    _DEVICE = "cpu"
    blobs = {}
    p = 0.2

    def run(case, layer, seed, *args):
        layer = _fill_by_name(layer, seed=seed).train()
        torch.manual_seed(2024)
        blobs[f"{case}.out"] = layer(*args).detach().cpu()

    shapes, lsi, S = levels([(6, 8)])
    grid = ts.DeformableTransformer.get_reference_points(shapes, torch.ones(2, 1, 2), "cpu")
    src, pos, depth = rnd(10, 2, S, 256), rnd(11, 2, S, 256), rnd(15, 2, S, 256)
    tgt, qpos = rnd(12, 2, 21, 256), rnd(13, 2, 21, 256)
    ref4 = urnd(14, 2, 21, 1, 4) * torch.tensor([1, 1, 0.4, 0.4])
    run("train_enc_layer", ts.DeformableTransformerEncoderLayer(256, 1024, p, "relu", 1, 8, 4), 2, src, pos, grid, shapes, lsi, None)
    run("train_dec_layer", ts.DeformableTransformerDecoderLayer(256, 1024, p, "relu", 1, 8, 4), 3,
        tgt, qpos, ref4, src, shapes, lsi, None)
    run("train_latefusion_layer", ts.DepthDeformableTransformerEncoderLayer(256, 1024, p, "relu", 1, 8, 4, True, True, True), 4,
        src, pos, None, shapes, grid, None, depth, shapes, lsi, None, None)
    run("train_fusion_v2_layer", ts.DeformableTransformerFusionLayerV2(256, 1024, p, "relu", 1, 8, 4), 5,
        src, pos, grid, depth, shapes, lsi, None)
    run("train_tqe_layer", tpp.TemporalQueryEncoderLayer(256, 1024, p, "relu", 8), 6, tgt[:1], rnd(16, 1, 55, 256))
    run("train_tdtd_layer", tpp.TemporalDeformableTransformerEncoderLayer(256, 1024, p, "relu", 1, 8, 4), 7,
        tgt, qpos, ref4[:, :, :, :2].contiguous(), src, shapes, lsi, None)
    cfg = {"MODEL": {"SparseRCNN": {"NHEADS": 8, "DROPOUT": 0.0, "DIM_FEEDFORWARD": 2048, "ACTIVATION": "relu",
                                    "HIDDEN_DIM": 256, "NUM_CLS": 1, "NUM_REG": 3, "NUM_HEADS": 6, "NUM_DYNAMIC": 2,
                                    "DIM_DYNAMIC": 64}, "ROI_BOX_HEAD": {"POOLER_RESOLUTION": 7}}}
    run("train_rcnn_head", RCNNHead(cfg, 256, 3, 1024, 8, p, "relu"), 8, rnd(17, 42, 256, 7, 7), rnd(18, 2, 21, 256))
    return blobs
