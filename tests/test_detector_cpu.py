"""CPU: detector-level host logic through ``build_model`` (the reference's entry point) with the
MSDA operator and RoIAlign routed to the CPU oracle."""
import pytest
import torch

from tests.test_models_golden import cpu_roi  # noqa: F401  (fixture)


def _small_clip(T, seed=0, H=64, W=96, C=4):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(T, C, H, W, generator=g)


@pytest.fixture()
def vodpp(cpu_msda, cpu_roi):  # noqa: F811
    from models import build_model
    from models.config import transvodpp_args
    from tests._param_fill import fill_params_by_name
    torch.manual_seed(0)
    model, criterion, post = build_model(transvodpp_args(num_ref_frames=2, device="cpu"))
    fill_params_by_name(model, seed=5)
    with torch.no_grad():                       # keep refined boxes inside the image
        for h in list(model.bbox_embed) + list(model.temp_bbox_embed_list):
            h.layers[-1].weight.mul_(0.2)
    return model.eval(), post


def test_build_model_surface(vodpp):
    model, post = vodpp
    keys = set(model.state_dict())
    for k in ("backbone.0.body.layer1.0.conv1.weight", "backbone.0.body.layer4.2.bn3.running_var",
              "backbone.0.body.layer1.0.downsample.0.weight", "backbone.0.body.bn1.weight",
              "depth_backbone.0.depth_backbone.downsample_layers_e.0.0.weight",
              "depth_backbone.0.depth_backbone.downsample_layers_e.3.1.weight",
              "input_proj.0.0.weight", "input_proj.0.1.bias", "input_proj_depth.0.0.weight",
              "query_embed.weight", "class_embed.5.weight", "bbox_embed.0.layers.2.bias",
              "temp_class_embed.weight", "temp_bbox_embed.layers.0.weight", "temp_class_embed_list.2.bias",
              "temp_bbox_embed_list.1.layers.1.weight", "transformer.level_embed",
              "transformer.depth_encoder_layer.cross_attn.sampling_offsets.weight",
              "transformer.decoder.bbox_embed.3.layers.0.weight"):
        assert k in keys, k
    assert model.state_dict()["query_embed.weight"].shape == (300, 512)
    assert model.state_dict()["input_proj.0.0.weight"].shape == (256, 2048, 1, 1)
    assert model.state_dict()["input_proj_depth.0.0.weight"].shape == (256, 128, 1, 1)
    assert "bbox" in post


def test_literal_forward_and_postprocess(vodpp):
    model, post = vodpp
    clip = _small_clip(3)
    from util.misc_multi import nested_tensor_from_tensor_list
    # the caller stacks the clip on the channel axis, [T*C,H,W]; the collate splits it into frames
    # (ref inference.py:883, util/misc_multi.py:319-340)
    samples = nested_tensor_from_tensor_list([clip.reshape(12, 64, 96)], split=True, channel_size=4)
    assert samples.tensors.shape == (3, 4, 64, 96) and not samples.mask.any()
    with torch.no_grad():
        out = model(samples)
    assert out["pred_logits"].shape == (1, 300, 3) and out["pred_boxes"].shape == (1, 300, 4)
    assert len(out["aux_outputs"]) == 2
    res = post["bbox"](out, torch.tensor([[64, 96]]))
    assert res[0]["scores"].shape == (100,) and res[0]["labels"].dtype == torch.int64
    # box index / label decomposition of the flat top-k (the "box indices" of the north star)
    prob = out["pred_logits"].sigmoid().view(1, -1)
    top = torch.topk(prob, 100, dim=1)[1]
    assert torch.equal(res[0]["labels"], (top % 3)[0])
    assert (res[0]["scores"][:-1] >= res[0]["scores"][1:]).all()


def test_all_current_runner_equals_reordered_literal_forward(vodpp):
    """ClipRunner output for frame t == the reference-style forward on [t, others in clip order]."""
    from models.clip_inference import ClipRunner
    from util.misc import NestedTensor
    model, _ = vodpp
    clip = _small_clip(3, seed=3)
    mask = torch.zeros(3, 64, 96, dtype=torch.bool)
    got = ClipRunner(model, micro_batch=2)(clip, mask)
    for t in range(3):
        order = [t] + [j for j in range(3) if j != t]
        with torch.no_grad():
            ref = model(NestedTensor(clip[order], mask[order]))
        assert torch.allclose(got["pred_logits"][t:t + 1], ref["pred_logits"], atol=2e-4), t
        assert torch.allclose(got["pred_boxes"][t:t + 1], ref["pred_boxes"], atol=2e-4), t


@pytest.mark.parametrize("fusion", ["Baseline", "LateFusion", "Encoder_CrossFusion", "Backbone_CrossFusion"])
def test_single_frame_configs_build_and_run(cpu_msda, fusion):
    from models import build_model
    from models.config import single_args
    from tests._param_fill import fill_params_by_name
    model, _, post = build_model(single_args(fusion, device="cpu"))
    fill_params_by_name(model, seed=6)
    model.eval()
    x = _small_clip(2, seed=1, C=4 if fusion != "Baseline" else 3)
    with torch.no_grad():
        out = model([x[0], x[1]])
    assert out["pred_logits"].shape == (2, 300, 3) and out["pred_boxes"].shape == (2, 300, 4)
    assert len(out["aux_outputs"]) == 5
    assert torch.isfinite(out["pred_logits"]).all()


def test_transvod_builds_and_runs(cpu_msda):
    from models import build_model
    from models.config import transvod_args
    from tests._param_fill import fill_params_by_name
    model, _, _ = build_model(transvod_args(num_ref_frames=2, device="cpu"))
    fill_params_by_name(model, seed=7)
    with torch.no_grad():
        out = model.eval()([_small_clip(3, seed=2, C=3).reshape(9, 64, 96)])
    assert out["pred_logits"].shape == (1, 300, 3)


def test_resnet50_body_keeps_torchvisions_checkpoint_contract():
    """torchvision is absent on both sides, so the ResNet-50 body cannot be pinned by a reference run; what is
    checked is torchvision's published contract that reference checkpoints are written against: 25 557 032 parameters
    (23 508 032 without the classifier), the block layout 3-4-6-3, the key names, FrozenBatchNorm2d buffers without
    num_batches_tracked, and stride 16 with the last stage dilated (DC5)."""
    import torch
    from models.backbone_scratch import FrozenBatchNorm2d
    from models.resnet import ResNet50
    m = ResNet50(FrozenBatchNorm2d, replace_stride_with_dilation=[False, False, True])
    sd = m.state_dict()
    n_params = sum(p.numel() for p in m.parameters())
    n_frozen = sum(b.weight.numel() + b.bias.numel() for b in m.modules() if isinstance(b, FrozenBatchNorm2d))   # affine terms live in buffers
    assert n_params + n_frozen == 25_557_032, n_params + n_frozen
    assert [len(getattr(m, f"layer{i}")) for i in (1, 2, 3, 4)] == [3, 4, 6, 3]
    for key in ("conv1.weight", "bn1.running_var", "layer1.0.downsample.0.weight", "layer1.0.downsample.1.running_mean",
                "layer3.5.conv3.weight", "layer4.2.bn3.bias", "fc.weight", "fc.bias"):
        assert key in sd, key
    assert not any(k.endswith("num_batches_tracked") for k in sd)
    assert sd["layer4.0.conv2.weight"].shape == (512, 512, 3, 3) and m.layer4[0].conv2.stride == (1, 1)
    assert m.layer4[1].conv2.dilation == (2, 2) and m.layer4[0].conv2.dilation == (1, 1)
    with torch.no_grad():
        x = m.stem(torch.zeros(1, 3, 64, 96))
        for stage in (m.layer1, m.layer2, m.layer3, m.layer4):
            x = stage(x)
    assert x.shape == (1, 2048, 4, 6)
