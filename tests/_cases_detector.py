"""Shared recipes for the DETECTOR-level golden vectors.

``run_detector_cases(ns)`` builds the single-frame Deformable-DETR (Late Fusion) and the TransVOD++ detector
(Late Fusion) from the module namespace ``ns`` around STUB backbones that return seeded feature maps (the real
backbones need torchvision / pretrained downloads on the reference side), fills every parameter by
state_dict name (tests/_param_fill.py), runs ``model(NestedTensor)`` and the post-processing the caller applies:

  * ``pred_logits`` / ``pred_boxes`` / ``aux_outputs``                       (deformable_detr_single.py:204-362,
                                                                              deformable_detr_multi_plusplus.py:210-342)
  * ``PostProcess`` scores / labels / boxes and the implied box indices       (deformable_detr_single.py:569-603)
  * the inference filter ``softmax(-1)[0][:, 1] > keep_prob``                 (inference.py:918-930)
  * the detector's ``state_dict`` keys and shapes (the checkpoint wire format; stub backbones
    hold no parameters, so these are the transformer, ``input_proj*``, heads and query embeddings)
  * every ``torch.topk`` call the forward makes (values + int64 indices, in call order): the temporal stage's
    ordered picks of k*R reference queries                                    (deformable_transformer_multi_plusplus.py:529,554,576)

tools/gen_golden_detector.py runs it with the REFERENCE's classes and stores tests/golden/detector.npz;
tests/test_detector_golden.py (CPU, oracle operators) and tests/test_models_gpu.py (HIP kernels) run it with this
repository's classes.  ``ns``: single, multipp (detector modules: DeformableDETR, PostProcess), ts, tpp
(transformer modules), NestedTensor, NestedTensorMulti (util.misc_multi), PositionEmbeddingSine.
"""
import torch
from torch import nn

from tests._param_fill import fill_params_by_name

KEEP_PROB = 0.3


def _rnd(seed, *shape, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


class StubJoiner(nn.Module):
    """Stands where Joiner(backbone, position_embedding) stands: ``self(samples) -> (features, pos, None, None)``
    (the fusion-backbone call shape, backbone_scratch.py:168-187) with one seeded stride-32 feature map; ``self[1]`` is
    the positional encoding (used by the detectors for extra levels)."""

    strides = [32]
    name = "resnet50"
    d_name = "dformer"

    def __init__(self, pos, NestedTensor, channels, seed, depth=False):
        super().__init__()
        self.pos, self.NT, self.channels, self.seed, self.depth = pos, NestedTensor, channels, seed, depth
        self.num_channels = [channels]

    def __getitem__(self, i):
        return (self, self.pos)[i]

    def forward(self, samples):
        x, m = samples.tensors, samples.mask
        h, w = -(-x.shape[2] // 32), -(-x.shape[3] // 32)
        feat = _rnd(self.seed, x.shape[0], self.channels, h, w).relu().to(x.device)
        mask = torch.nn.functional.interpolate(m[None].float(), size=(h, w)).to(torch.bool)[0]
        nt = self.NT(feat, mask)
        pos = [self.pos(nt).to(feat.dtype)]
        return ([nt], pos) if self.depth else ([nt], pos, None, None)


class TopkRecorder:
    """Records every torch.topk call made inside the ``with`` block (values, indices)."""

    def __enter__(self):
        self.calls, self._orig = [], torch.topk

        def topk(*a, **k):
            out = self._orig(*a, **k)
            self.calls.append((out[0].detach().cpu().clone(), out[1].detach().cpu().clone()))
            return out

        torch.topk = topk
        return self

    def __exit__(self, *exc):
        torch.topk = self._orig


def round_picks(calls, R):
    """The three rounds' (values, indices): the reference calls torch.topk once per round (80 R, 50 R, 30 R picks of the same
    scores); an implementation may select once with the largest k and take sorted prefixes - one recorded call."""
    if len(calls) == 1:
        vals, idx = calls[0]
        assert vals.shape[1] == 80 * R
        return [(vals[:, :k * R], idx[:, :k * R]) for k in (80, 50, 30)]
    assert len(calls) == 3, "the temporal stage makes three top-k picks"
    return calls


def run_detector_cases(ns, device="cpu"):
    NestedTensor, PE = ns.NestedTensor, ns.PositionEmbeddingSine
    NTM = getattr(ns, "NestedTensorMulti", NestedTensor)     # util.misc_multi's own NestedTensor class
    blobs = {}

    def put(case, **tensors):
        for k, v in tensors.items():
            blobs[f"{case}.{k}"] = v.detach().cpu()

    def put_state_dict(case, model):
        # the checkpoint wire format: every state_dict key with its shape, as JSON bytes
        import json
        doc = json.dumps({k: list(v.shape) for k, v in model.state_dict().items()}, sort_keys=True)
        blobs[f"{case}.state_dict_json"] = torch.frombuffer(bytearray(doc.encode()), dtype=torch.uint8).clone()

    def post(case, det_mod, out, sizes, keep_prob=KEEP_PROB):
        res = det_mod.PostProcess()(out, sizes)
        logits = out["pred_logits"]
        # the indices PostProcess gathers the boxes with (it does not return them): same rule, same call
        idx = torch.topk(logits.sigmoid().view(logits.shape[0], -1), 100, dim=1)[1]
        put(case, pp_scores=torch.stack([r["scores"] for r in res]), pp_labels=torch.stack([r["labels"] for r in res]),
            pp_boxes=torch.stack([r["boxes"] for r in res]), pp_box_idx=idx // logits.shape[2])
        probas = logits.softmax(-1)[0]
        put(case, keep_probas=probas[:, 1], keep_mask=probas[:, 1] > keep_prob, keep_prob=torch.tensor(keep_prob))

    dtype_str = "DepthDeform_latefusion_dformer"
    H, W = 160, 256                                        # stride-32 map: 5 x 8

    # ---- RGB backbone wrapper (a14): FusionBackboneBase.forward + FrozenBatchNorm2d + Joiner around a small body -----
    # (the real body is torchvision's ResNet-50, absent on both sides; what is pinned is everything the reference's own
    # file does: RGB channel slice of the RGB-D input, stem / layer call order, FrozenBatchNorm2d arithmetic, the mask
    # resized to every returned level, the returned-level dictionaries for one and for three levels, positions per level)
    if hasattr(ns, "bsc"):
        FBN = ns.bsc.FrozenBatchNorm2d

        def block(ci, co, k, stride, dilation=1):
            return nn.Sequential(nn.Conv2d(ci, co, k, stride, padding=dilation * (k // 2), dilation=dilation, bias=False), FBN(co), nn.ReLU())

        class Body(nn.Module):
            def __init__(self):
                super().__init__()
                self.conv1, self.bn1, self.relu = nn.Conv2d(3, 8, 7, 2, 3, bias=False), FBN(8), nn.ReLU()
                self.maxpool = nn.MaxPool2d(3, 2, 1)
                self.layer1, self.layer2 = block(8, 16, 1, 1), block(16, 24, 3, 2)
                self.layer3, self.layer4 = block(24, 32, 3, 2), block(32, 40, 3, 1, dilation=2)

        for interm in (False, True):
            pe_b = PE(128, normalize=True)
            base = ns.bsc.FusionBackboneBase("resnet50", "resnet18", Body(), None, pe_b, True, interm, dtype_str, [3], 256, False)
            joiner = ns.bsc.Joiner(base, pe_b).eval()
            fill_params_by_name(joiner, seed=71)
            joiner = joiner.to(device)
            xb = _rnd(72, 2, 4, 96, 128).to(device)
            mb = torch.zeros(2, 96, 128, dtype=torch.bool)
            mb[1, 70:, :] = True
            mb[1, :, 100:] = True
            feats, pos, _, _ = joiner(NestedTensor(xb, mb.to(device)))
            case = f"backbone_wrapper_interm{int(interm)}"
            put(case, n_levels=torch.tensor(len(feats)))
            for i, (f, p_) in enumerate(zip(feats, pos)):
                put(case, **{f"feat{i}": f.tensors, f"mask{i}": f.mask, f"pos{i}": p_})
            put_state_dict(case, joiner)

    def masks(n):
        m = torch.zeros(n, H, W, dtype=torch.bool)
        if n > 1:
            m[1, :, 200:] = True                           # frame 1 is padded on the right
        return m.to(device)

    # ---- single-frame Deformable-DETR + Late Fusion (a16, a18) ---------------------------------------------
    B, Q = 2, 120
    tr = ns.ts.DeformableTransformer(d_model=256, nhead=8, num_encoder_layers=2, num_decoder_layers=2,
                                     dim_feedforward=1024, dropout=0.1, activation="relu", return_intermediate_dec=True,
                                     num_feature_levels=1, dec_n_points=4, enc_n_points=4, two_stage=False,
                                     two_stage_num_proposals=Q, use_depth=True, depth_type=dtype_str, dpth_n_points=4)
    pe = PE(128, normalize=True)
    det = ns.single.DeformableDETR(StubJoiner(pe, NestedTensor, 2048, 301), StubJoiner(pe, NestedTensor, 128, 302, depth=True),
                                   tr, num_classes=3, num_queries=Q, num_feature_levels=1, aux_loss=True,
                                   with_box_refine=True, two_stage=False, use_depth=True, depth_type=dtype_str).eval()
    fill_params_by_name(det, seed=41)
    put_state_dict("det_single", det)
    det = det.to(device)
    x = _rnd(303, B, 4, H, W).to(device)
    with TopkRecorder():
        out = det(NestedTensor(x, masks(B)))
    put("det_single", pred_logits=out["pred_logits"], pred_boxes=out["pred_boxes"],
        aux0_logits=out["aux_outputs"][0]["pred_logits"], aux0_boxes=out["aux_outputs"][0]["pred_boxes"])
    post("det_single", ns.single, out, torch.as_tensor([[480, 640], [300, 400]], device=device))

    # ---- TransVOD++ + Late Fusion: clip of 1 + R frames, output for frame 0 (a16, a11, a18) ------------------
    R, Q = 2, 90
    tr = ns.tpp.DeformableTransformer(d_model=256, nhead=8, num_encoder_layers=2, num_decoder_layers=2,
                                      dim_feedforward=1024, dropout=0.1, activation="relu", return_intermediate_dec=True,
                                      num_feature_levels=1, dec_n_points=4, enc_n_points=4, two_stage=False,
                                      two_stage_num_proposals=Q, num_query=Q, n_temporal_decoder_layers=1,
                                      num_ref_frames=R, fixed_pretrained_model=False, args=None, use_depth=True,
                                      depth_type=dtype_str, dpth_n_points=4)
    det = ns.multipp.DeformableDETR(StubJoiner(pe, NTM, 2048, 311), StubJoiner(pe, NTM, 128, 312, depth=True),
                                    tr, num_classes=3, num_queries=Q, num_feature_levels=1, num_ref_frames=R,
                                    aux_loss=True, with_box_refine=True, two_stage=False, use_depth=True,
                                    depth_type=dtype_str).eval()
    fill_params_by_name(det, seed=51)
    put_state_dict("det_multipp", det)
    det = det.to(device)
    x = _rnd(313, R + 1, 4, H, W).to(device)
    with TopkRecorder() as rec:
        out = det(NTM(x, torch.zeros(R + 1, H, W, dtype=torch.bool, device=device)))
    put("det_multipp", pred_logits=out["pred_logits"], pred_boxes=out["pred_boxes"],
        aux0_logits=out["aux_outputs"][0]["pred_logits"], aux1_boxes=out["aux_outputs"][1]["pred_boxes"])
    for i, (vals, idx) in enumerate(round_picks(rec.calls, R)):
        put("det_multipp", **{f"topk{i}_values": vals, f"topk{i}_idx": idx})
    post("det_multipp", ns.multipp, out, torch.as_tensor([[480, 640]], device=device))

    # ---- TransVOD++ RGB (``--fusion_type Baseline``, configs/training/TransVOD++.sh; BASELINE.json configs[3]): the
    # 3-channel branch of the multi++ detector - no depth backbone, no input_proj_depth, no fusion layer (a16, a11) ----
    R, Q = 3, 90
    tr = ns.tpp.DeformableTransformer(d_model=256, nhead=8, num_encoder_layers=2, num_decoder_layers=2,
                                      dim_feedforward=1024, dropout=0.1, activation="relu", return_intermediate_dec=True,
                                      num_feature_levels=1, dec_n_points=4, enc_n_points=4, two_stage=False,
                                      two_stage_num_proposals=Q, num_query=Q, n_temporal_decoder_layers=1,
                                      num_ref_frames=R, fixed_pretrained_model=False, args=None, use_depth=False,
                                      depth_type="Baseline_rgb", dpth_n_points=4)
    det = ns.multipp.DeformableDETR(StubJoiner(pe, NTM, 2048, 321), None, tr, num_classes=3, num_queries=Q,
                                    num_feature_levels=1, num_ref_frames=R, aux_loss=True, with_box_refine=True,
                                    two_stage=False, use_depth=False, depth_type="Baseline_rgb").eval()
    fill_params_by_name(det, seed=61)
    put_state_dict("det_multipp_rgb", det)
    det = det.to(device)
    x = _rnd(323, R + 1, 3, H, W).to(device)
    with TopkRecorder() as rec:
        out = det(NTM(x, torch.zeros(R + 1, H, W, dtype=torch.bool, device=device)))
    put("det_multipp_rgb", pred_logits=out["pred_logits"], pred_boxes=out["pred_boxes"],
        aux0_logits=out["aux_outputs"][0]["pred_logits"], aux1_boxes=out["aux_outputs"][1]["pred_boxes"])
    for i, (vals, idx) in enumerate(round_picks(rec.calls, R)):
        put("det_multipp_rgb", **{f"topk{i}_values": vals, f"topk{i}_idx": idx})
    post("det_multipp_rgb", ns.multipp, out, torch.as_tensor([[360, 480]], device=device), keep_prob=0.5)

    # ---- TransVOD (``--dataset_file vid_multi``, models/deformable_detr_multi.py:45-311 around
    # deformable_transformer_multi.py:193-378): one pair of temporal heads, the transformer gets only class_embed[-1],
    # the three picks rank ``sigmoid(logits)[:, :, :-1]`` - Late Fusion and RGB ---------------------------------------
    if hasattr(ns, "multi"):
        for case, use_depth, seed, R, Q, keep in (("det_multi", True, 81, 2, 90, 0.02), ("det_multi_rgb", False, 91, 3, 60, 0.3)):
            dt = dtype_str if use_depth else "Baseline_rgb"
            tr = ns.tm.DeformableTransformer(d_model=256, nhead=8, num_encoder_layers=2, num_decoder_layers=2,
                                             dim_feedforward=1024, dropout=0.1, activation="relu", return_intermediate_dec=True,
                                             num_feature_levels=1, dec_n_points=4, enc_n_points=4, two_stage=False,
                                             two_stage_num_proposals=Q, n_temporal_decoder_layers=1, num_ref_frames=R,
                                             fixed_pretrained_model=False, args=None, use_depth=use_depth, depth_type=dt,
                                             dpth_n_points=4)
            det = ns.multi.DeformableDETR(StubJoiner(pe, NTM, 2048, seed + 1),
                                          StubJoiner(pe, NTM, 128, seed + 2, depth=True) if use_depth else None, tr,
                                          num_classes=3, num_queries=Q, num_feature_levels=1, num_ref_frames=R, aux_loss=True,
                                          with_box_refine=True, two_stage=False, use_depth=use_depth, depth_type=dt).eval()
            fill_params_by_name(det, seed=seed)
            put_state_dict(case, det)
            det = det.to(device)
            x = _rnd(seed + 3, R + 1, 4 if use_depth else 3, H, W).to(device)
            with TopkRecorder() as rec:
                out = det(NTM(x, torch.zeros(R + 1, H, W, dtype=torch.bool, device=device)))
            put(case, pred_logits=out["pred_logits"], pred_boxes=out["pred_boxes"])
            for i, (vals, idx) in enumerate(round_picks(rec.calls, R)):
                put(case, **{f"topk{i}_values": vals, f"topk{i}_idx": idx})
            post(case, ns.multi, out, torch.as_tensor([[480, 640]], device=device), keep_prob=keep)
    return blobs


def compare_indices(ref_idx, got_idx, ref_scores, margin):
    """Ordered index tensors [B,k] from a top-k: equal wherever the reference score at that rank is separated from its
    neighbours by more than ``margin`` (ties and near-ties may legitimately swap under fp32 rounding differences).
    Returns (number compared, number of mismatches among them)."""
    s = ref_scores
    gap_prev = torch.cat([torch.full_like(s[:, :1], float("inf")), (s[:, :-1] - s[:, 1:]).abs()], 1)
    gap_next = torch.cat([(s[:, :-1] - s[:, 1:]).abs(), torch.full_like(s[:, :1], float("inf"))], 1)
    clear = (gap_prev > margin) & (gap_next > margin)
    return int(clear.sum()), int((ref_idx[clear] != got_idx[clear]).sum())
