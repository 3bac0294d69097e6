"""Argument namespaces equal to what the reference's CLIs hand to ``build_model``: the argparse
defaults of main.py:31-193 / main_multi.py:28-177 overlaid with the flags of the shipped configurations
(configs/training/*.sh).  Only the fields ``build_model`` reads are listed.  Pinned field by field to the
reference's own parsers and scripts by tests/golden/args.json (tools/gen_golden_args.py runs
``get_args_parser()`` on every script's command line; tests/test_args_contract.py).
"""
from types import SimpleNamespace

# parser defaults overlaid with what EVERY shipped configuration passes:
# --backbone resnet50 --num_feature_levels 1 --num_queries 300 --dilation --with_box_refine --num_classes 3 --dropout 0.2
_DEFAULTS = dict(
    lr_backbone=2e-5, backbone="resnet50", dilation=True, position_embedding="sine", masks=False,
    num_feature_levels=1, enc_layers=6, dec_layers=6, dim_feedforward=1024, hidden_dim=256, dropout=0.2,
    nheads=8, num_queries=300, dec_n_points=4, enc_n_points=4, dpth_n_points=4, two_stage=False,
    with_box_refine=True, aux_loss=True, num_classes=3, device="cuda", frozen_weights=None,
    set_cost_class=2, set_cost_bbox=5, set_cost_giou=2, mask_loss_coef=1, dice_loss_coef=1, cls_loss_coef=2,
    bbox_loss_coef=5, giou_loss_coef=2, focal_alpha=0.25, dformer_weights=None, dformer_backbone=False,
    use_depth=False, fusion_type="Baseline", depth_type="Baseline_rgb", num_ref_frames=3,
    n_temporal_decoder_layers=1, fixed_pretrained_model=False, dataset_file="vid_single")

# what each script adds to that (paths - --dformer_weights, --resume ... - are the caller's business)
SHIPPED = {
    "Baseline.sh": dict(dataset_file="vid_single"),
    "LateFusion.sh": dict(dataset_file="vid_single", use_depth=True, dformer_backbone=True, fusion_type="LateFusion"),
    "Encoder_CrossFusion.sh": dict(dataset_file="vid_single", use_depth=True, dformer_backbone=True,
                                   fusion_type="Encoder_CrossFusion"),
    "Backbone_CrossFusion.sh": dict(dataset_file="vid_single", use_depth=True, fusion_type="Backbone_CrossFusion"),
    "TransVOD++.sh": dict(dataset_file="vid_multi_plusplus", num_ref_frames=4),
    "TransVOD++_withdepth.sh": dict(dataset_file="vid_multi_plusplus", num_ref_frames=4, use_depth=True,
                                    dformer_backbone=True, fusion_type="LateFusion"),
}


def make_args(**overrides):
    a = dict(_DEFAULTS)
    a.update(overrides)
    return SimpleNamespace(**a)


def config_args(script, **kw):
    """The namespace of one shipped configuration, e.g. ``config_args("TransVOD++_withdepth.sh", num_ref_frames=31)``."""
    return make_args(**{**SHIPPED[script], **kw})


def single_args(fusion_type="Baseline", **kw):
    """configs/training/{Baseline,LateFusion,Encoder_CrossFusion,Backbone_CrossFusion}.sh"""
    return config_args(f"{fusion_type}.sh", **kw)


def transvodpp_args(fusion_type="LateFusion", num_ref_frames=4, **kw):
    """configs/training/TransVOD++{,_withdepth}.sh"""
    script = "TransVOD++.sh" if fusion_type == "Baseline" else "TransVOD++_withdepth.sh"
    return config_args(script, **{"fusion_type": fusion_type, "num_ref_frames": num_ref_frames, **kw})


def transvod_args(fusion_type="Baseline", num_ref_frames=4, **kw):
    """TransVOD (``--dataset_file vid_multi``; no shipped script)"""
    depth = fusion_type != "Baseline"
    return make_args(dataset_file="vid_multi", fusion_type=fusion_type, use_depth=depth, dformer_backbone=depth,
                     num_ref_frames=num_ref_frames, **kw)
