"""Shared recipe of the VIDEO-inference fixture (round 3): a synthetic video of n frames, each frame
detected with the reference frames the reference's caller would give it (inference.py:721-794: the window
[t - R, t + R] without t, first R entries, repeated when short).

``build_detector(ns, R, depth)`` builds the TransVOD++ detector of the module namespace ``ns`` around stub backbones whose
feature maps DEPEND ON THE FRAME (a seeded linear map of the 32x-pooled pixels plus a seeded positional pattern; the real
backbones need torchvision on the reference side), weights filled by state_dict name.  tools/gen_golden_stream.py runs the
reference's own caller steps (``get_image_and_reference_clips`` -> ``util.misc_multi.nested_tensor_from_tensor_list`` ->
``DeformableDETR.forward``) per frame and stores the outputs; tests/test_stream.py feeds the same frames to
``models.clip_inference.VideoStream`` in blocks."""
import torch
import torch.nn.functional as F
from torch import nn

from tests._param_fill import fill_params_by_name

VIDEOS = {"long_rgbd": dict(n=9, R=3, depth=True, seed=400), "short_rgb": dict(n=3, R=4, depth=False, seed=410),
          "exact_rgbd": dict(n=4, R=3, depth=True, seed=420)}
H, W, Q = 96, 160, 90


def _rnd(seed, *shape):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


def video_frames(name):
    v = VIDEOS[name]
    return _rnd(v["seed"], v["n"], 4 if v["depth"] else 3, H, W)


class ContentJoiner(nn.Module):
    """Stands where Joiner(backbone, position_embedding) stands; features are a function of the frame."""
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
        pooled = F.adaptive_avg_pool2d(x, (h, w)) * 32.0
        proj = _rnd(self.seed, self.channels, x.shape[1]).to(x.device)
        pattern = _rnd(self.seed + 1, self.channels, h, w).to(x.device)
        feat = (torch.einsum("oc,nchw->nohw", proj, pooled) + 0.5 * pattern).relu()
        mask = F.interpolate(m[None].float(), size=(h, w)).to(torch.bool)[0]
        nt = self.NT(feat, mask)
        pos = [self.pos(nt).to(feat.dtype)]
        return ([nt], pos) if self.depth else ([nt], pos, None, None)


def build_detector(ns, R, depth, seed=91):
    dtype_str = "DepthDeform_latefusion_dformer" if depth else "Baseline_rgb"
    tr = ns.tpp.DeformableTransformer(d_model=256, nhead=8, num_encoder_layers=1, num_decoder_layers=2,
                                      dim_feedforward=1024, dropout=0.1, activation="relu", return_intermediate_dec=True,
                                      num_feature_levels=1, dec_n_points=4, enc_n_points=4, two_stage=False,
                                      two_stage_num_proposals=Q, num_query=Q, n_temporal_decoder_layers=1,
                                      num_ref_frames=R, fixed_pretrained_model=False, args=None, use_depth=depth,
                                      depth_type=dtype_str, dpth_n_points=4)
    pe = ns.PositionEmbeddingSine(128, normalize=True)
    NTM = ns.NestedTensorMulti
    det = ns.multipp.DeformableDETR(ContentJoiner(pe, NTM, 2048, 501),
                                    ContentJoiner(pe, NTM, 128, 502, depth=True) if depth else None,
                                    tr, num_classes=3, num_queries=Q, num_feature_levels=1, num_ref_frames=R,
                                    aux_loss=True, with_box_refine=True, two_stage=False, use_depth=depth,
                                    depth_type=dtype_str).eval()
    fill_params_by_name(det, seed=seed)
    return det
