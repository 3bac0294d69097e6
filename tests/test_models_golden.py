"""CPU: this repository's host-side modules (MSDeformAttn, fusion layers, encoder / decoder layers,
TQE, RCNNHead, DFormer backbone, positional encoding, the three transformers) against outputs of
the REFERENCE's modules on the same seeded inputs and name-keyed weights (tests/_cases.py,
tools/gen_golden_models.py).  The MSDA operator is the CPU oracle here (no GPU); RoIAlign is the
oracle restatement on both sides.  test_models_gpu.py repeats the comparison with the HIP kernels.
"""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch


def my_namespace():
    import models.deformable_transformer_multi as tm
    import models.deformable_transformer_multi_plusplus as tpp
    import models.deformable_transformer_single as ts
    import models.dformer_backbone as dfb
    import models.dformer_crossfusion_backbone as dcf
    from models.ops.modules import MSDeformAttn
    from models.position_encoding import PositionEmbeddingSine
    from models.sparse_roi_head.head import RCNNHead
    from util.misc import NestedTensor, inverse_sigmoid
    return SimpleNamespace(MSDeformAttn=MSDeformAttn, ts=ts, tpp=tpp, tm=tm, RCNNHead=RCNNHead, dfb=dfb, dcf=dcf,
                           PositionEmbeddingSine=PositionEmbeddingSine, NestedTensor=NestedTensor,
                           inverse_sigmoid=inverse_sigmoid)


@pytest.fixture()
def cpu_roi(oracle, monkeypatch):
    from dfx import ops

    def roi_align(inp, rois, output_size, spatial_scale, sampling_ratio, aligned=True, channels_last=False):
        size = output_size if isinstance(output_size, int) else output_size[0]
        if channels_last:
            out = oracle.roi_align(inp.permute(0, 3, 1, 2).contiguous(), rois, size, spatial_scale, sampling_ratio, aligned)
            return out.flatten(2).transpose(1, 2).contiguous()
        return oracle.roi_align(inp, rois, size, spatial_scale, sampling_ratio, aligned)

    monkeypatch.setattr(ops, "roi_align", roi_align)


@pytest.fixture(scope="module")
def golden(golden_dir):
    return np.load(os.path.join(golden_dir, "models.npz"))


def test_every_block_matches_the_reference(golden, cpu_msda, cpu_roi):
    from tests._cases import run_cases
    with torch.no_grad():
        got = run_cases(my_namespace())
    assert set(got) == set(golden.files)
    worst = {}
    for key in sorted(got):
        ref = torch.from_numpy(golden[key])
        out = got[key]
        assert out.shape == ref.shape, key
        if ref.dtype == torch.bool:
            assert torch.equal(out, ref), key
            continue
        err = (out.float() - ref.float()).abs().max().item()
        worst[key] = err
        # attention outputs within 1e-3 (north star); these CPU paths agree far tighter
        assert err < 2e-4, f"{key}: max abs err {err:.3e}"
    print("max abs err per block:", {k: f"{v:.1e}" for k, v in worst.items()})


def test_transformer_state_dict_keys_match_survey_contract():
    """TransVOD++ Late Fusion transformer = 53.405 M parameters, named groups."""
    from models.deformable_transformer_multi_plusplus import DeformableTransformer
    t = DeformableTransformer(num_feature_levels=1, depth_type="DepthDeform_latefusion_dformer", use_depth=True,
                              num_ref_frames=4, return_intermediate_dec=True)
    assert abs(sum(p.numel() for p in t.parameters()) / 1e6 - 53.405) < 1e-3
    keys = set(t.state_dict())
    for k in ("level_embed", "reference_points.weight",
              "encoder.layers.0.self_attn.sampling_offsets.weight", "encoder.layers.5.norm2.bias",
              "depth_encoder_layer.cross_attn.value_proj.weight", "depth_encoder_layer.norm3.weight",
              "depth_encoder_layer.depth_scale_adapt.weight", "depth_encoder_layer.norm_depth_scale.bias",
              "depth_encoder_layer.cross_scale_adapt.bias",
              "decoder.layers.0.cross_attn.output_proj.bias", "decoder.layers.5.self_attn.in_proj_weight",
              "temporal_query_layer1.cross_attn.in_proj_bias", "temporal_query_layer3.norm3.weight",
              "dynamic_layer_for_current_query1.inst_interact.dynamic_layer.weight",
              "dynamic_layer_for_current_query3.inst_interact.out_layer.bias",
              "temporal_decoder1.layers.0.cross_attn.sampling_offsets.bias", "temporal_decoder3.layers.0.norm3.bias"):
        assert k in keys, k
    assert t.state_dict()["dynamic_layer_for_current_query1.inst_interact.dynamic_layer.weight"].shape == (32768, 256)
    assert t.state_dict()["dynamic_layer_for_current_query1.inst_interact.out_layer.weight"].shape == (256, 12544)


def test_encoder_cf_uses_norm2_in_fusion_layers():
    from models.deformable_transformer_single import DeformableTransformer
    t = DeformableTransformer(num_feature_levels=1, depth_type="DepthDeform_encoder_cf_dformer", use_depth=True)
    keys = set(t.state_dict())
    assert "encoder.fusion_layers.3.norm2.weight" in keys and "encoder.fusion_layers.0.norm3.weight" not in keys
    assert "encoder.fusion_layers.0.depth_scale_adapt.weight" in keys


def test_train_mode_blocks_match_the_reference_under_a_fixed_seed(golden_dir, cpu_msda):
    """The sub-layer Dropouts (p = 0.2 in every shipped config) are applied where the reference applies them - to the same
    tensor, in the same order - so that a train-mode forward draws the same masks from the same generator state
    (tests/_cases.py:run_train_cases; fixture from the reference's modules, tools/gen_golden_models.py)."""
    from tests._cases import run_train_cases
    ref = np.load(os.path.join(golden_dir, "train_mode.npz"))
    with torch.no_grad():
        got = run_train_cases(my_namespace())
    assert set(got) == set(ref.files) and len(got) == 7
    for key in sorted(got):
        err = (got[key] - torch.from_numpy(ref[key])).abs().max().item()
        assert err < 2e-4, f"{key}: max abs err {err:.3e}"


def test_train_mode_differs_from_eval():
    """... and a dropped Dropout would show: with an active Dropout the fused-epilogue helper must not return the eval result."""
    from models.transformer_layers import _linear_norm_add
    x = torch.randn(2, 5, 256)
    lin, norm, drop = torch.nn.Linear(256, 256), torch.nn.LayerNorm(256), torch.nn.Dropout(0.5)
    torch.manual_seed(0)
    a = _linear_norm_add(lin, x, norm, x, dropout=drop.train())
    b = _linear_norm_add(lin, x, norm, x, dropout=drop.eval())
    assert not torch.allclose(a, b) and torch.allclose(b, norm(x + lin(x)), atol=1e-6)
