"""GPU parity of the whole host-side stack on the HIP kernels: every block of tests/_cases.py run
on cuda:0 (fused MSDA front end, HIP RoIAlign) against the reference's outputs (models.npz), the
detector through build_model against the same detector on CPU with the oracle operators, and the
box indices of the post-processor."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def golden(golden_dir):
    return np.load(os.path.join(golden_dir, "models.npz"))


def test_every_block_matches_the_reference_on_gpu(golden):
    from dfx import _lib
    from tests._cases import pad_masks, run_cases
    from tests.test_models_golden import my_namespace
    _lib.load()
    with torch.no_grad():
        got = run_cases(my_namespace(), device="cuda")
    assert set(got) == set(golden.files)
    for key in sorted(got):
        ref = torch.from_numpy(golden[key])
        out = got[key]
        if ref.dtype == torch.bool:
            assert torch.equal(out, ref), key
            continue
        diff = (out.float() - ref.float()).abs()
        if key in ("dformer.pos", "pos_sine.out"):
            # fully padded rows/columns normalise by eps: sin/cos of ~1e6, where CPU and GPU range
            # reduction legitimately differ; only positions of valid pixels are meaningful
            valid = ~(torch.from_numpy(golden["dformer.feat_mask"]) if key == "dformer.pos" else pad_masks(23, 2, 7, 9).cpu())
            diff = diff * valid[:, None].to(diff.dtype)
        if key == "fuse_layers.out":
            # same effect one step later: queries AT padded pixels carry those positional values
            valid = ~pad_masks(21, 1, 5, 7).cpu()
            valid[0, :, 6:] = False
            diff = diff * valid[:, None].to(diff.dtype)
        err = diff.max().item()
        assert err < 1e-3, f"{key}: max abs err {err:.3e} (north-star bound 1e-3)"
        assert err < 2e-4, f"{key}: max abs err {err:.3e}"


def _clip(T, seed, H=64, W=96):
    return torch.randn(T, 4, H, W, generator=torch.Generator().manual_seed(seed))


def _build(device):
    from models import build_model
    from models.config import transvodpp_args
    from tests._param_fill import fill_params_by_name
    model, _, post = build_model(transvodpp_args(num_ref_frames=3, device=device))
    fill_params_by_name(model, seed=5)
    with torch.no_grad():
        for h in list(model.bbox_embed) + list(model.temp_bbox_embed_list):
            h.layers[-1].weight.mul_(0.2)
    return model.eval(), post


def test_detector_gpu_vs_cpu_oracle_and_box_indices(oracle, monkeypatch):
    """TransVOD++ Late Fusion through build_model: HIP path vs the same host code on CPU with the
    oracle operators; box indices of PostProcess must be identical wherever the score margin
    between neighbours in the ranking exceeds the numerical noise."""
    from models.clip_inference import ClipRunner
    clip = _clip(4, 21)
    gpu_model, post = _build("cuda")
    gpu_model = gpu_model.cuda()
    got = ClipRunner(gpu_model, micro_batch=2)(clip.cuda())

    import models.ops.functions.ms_deform_attn_func as f
    from dfx import ops
    from tests.test_clip_shard_gloo import _patch_cpu_ops
    saved = (f.MSDeformAttnFunction, ops.roi_align)
    try:
        _patch_cpu_ops()
        cpu_model, _ = _build("cpu")
        want = ClipRunner(cpu_model, micro_batch=2)(clip)
    finally:
        f.MSDeformAttnFunction, ops.roi_align = saved
    lg, bx = got["pred_logits"].cpu(), got["pred_boxes"].cpu()
    assert (lg - want["pred_logits"]).abs().max() < 1e-3
    assert (bx - want["pred_boxes"]).abs().max() < 1e-3
    sizes = torch.tensor([[64, 96]] * 4)
    res_g = post["bbox"]({"pred_logits": lg, "pred_boxes": bx}, sizes)
    res_c = post["bbox"](want, sizes)
    for rg, rc in zip(res_g, res_c):
        margin = (rc["scores"][:-1] - rc["scores"][1:]).abs()
        safe = torch.ones(100, dtype=torch.bool)
        safe[:-1] &= margin > 1e-5
        safe[1:] &= margin > 1e-5
        prob = want["pred_logits"].sigmoid()
        assert torch.equal(rg["labels"][safe], rc["labels"][safe])
        assert torch.allclose(rg["boxes"][safe], rc["boxes"][safe], atol=0.2)
        assert set(rg["labels"].tolist()) <= {0, 1, 2} and prob.shape[-1] == 3
    # temporal top-k picks (k*R reference queries): identical index sets
    for pg, pc in zip(got["topk"], want["topk"]):          # 3 rounds, each [F, k*R]
        for a, b in zip(pg.cpu(), pc):
            assert set(a.tolist()) == set(b.tolist())


@pytest.mark.parametrize("H,W,T", [
    (70, 118, 3),        # feature map 5 x 8: H*W not a multiple of 4 -> the library-convolution fall-backs of the backbone
    (1216, 1600, 2),     # feature map 76 x 100 = 7600 tokens: too large for the level-in-LDS kernel -> wave-per-query kernel
])
@pytest.mark.timeout(900)
def test_detector_other_resolutions_gpu_vs_cpu_oracle(H, W, T):
    """The same GPU-vs-CPU-oracle comparison at resolutions that leave the fast paths of the default
    800 x 1333 configuration (shape guards in models/resnet.py, dfx/ops.py:level_supported)."""
    from models.clip_inference import ClipRunner
    clip = _clip(T, 31, H, W)
    gpu_model, _ = _build("cuda")
    got = ClipRunner(gpu_model.cuda(), micro_batch=T)(clip.cuda())
    import models.ops.functions.ms_deform_attn_func as f
    from dfx import ops
    from tests.test_clip_shard_gloo import _patch_cpu_ops
    saved = (f.MSDeformAttnFunction, ops.roi_align)
    try:
        _patch_cpu_ops()
        cpu_model, _ = _build("cpu")
        want = ClipRunner(cpu_model, micro_batch=T)(clip)
    finally:
        f.MSDeformAttnFunction, ops.roi_align = saved
    assert (got["pred_logits"].cpu() - want["pred_logits"]).abs().max() < 1e-3
    assert (got["pred_boxes"].cpu() - want["pred_boxes"]).abs().max() < 1e-3


def test_roi_align_kernels_match_oracle(oracle):
    from dfx import ops
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, 256, 13, 21, generator=g)
    rois = torch.tensor([[0, 10., 12., 200., 150.], [1, -20., -5., 100., 400.], [0, 300., 100., 340., 140.],
                         [1, 50., 60., 50.5, 60.5], [0, 0., 0., 672., 416.]])
    want = oracle.roi_align(x, rois, 7, 1 / 32, 2, True)
    got = ops.roi_align(x.cuda(), rois.cuda(), 7, 1 / 32, 2, True).cpu()
    assert torch.allclose(got, want, atol=1e-5)
    tok = ops.roi_align(x.permute(0, 2, 3, 1).contiguous().cuda(), rois.cuda(), 7, 1 / 32, 2, True, channels_last=True)
    assert torch.allclose(tok.transpose(1, 2).reshape(5, 256, 7, 7).cpu(), want, atol=1e-5)


def test_fused_backbone_matches_reference_formulation():
    """ResNet-50 DC5: frozen BN folded into the convolutions + fused bias/residual/ReLU kernel vs the
    reference's conv -> FrozenBatchNorm2d -> ReLU op sequence (backbone_scratch.py:102-141)."""
    from models.backbone_scratch import build_backbone_fromscratch
    from models.config import single_args
    from models.fused import enable_fused_inference
    from tests._param_fill import fill_params_by_name
    from util.misc import NestedTensor
    args = single_args("Baseline")
    args.depth_type = "Baseline_rgb"
    bb = fill_params_by_name(build_backbone_fromscratch(args), seed=3).cuda().eval()
    x = torch.randn(2, 3, 96, 160, generator=torch.Generator().manual_seed(1)).cuda()
    nt = NestedTensor(x, torch.zeros(2, 96, 160, dtype=torch.bool, device="cuda"))
    with torch.no_grad():
        plain = bb(nt)[0][0].tensors
        assert enable_fused_inference(bb) >= 1
        fused = bb(nt)[0][0].tensors
    assert plain.shape == fused.shape == (2, 2048, 6, 10)
    scale = plain.abs().max().item()
    assert (plain - fused).abs().max().item() < 1e-4 * max(scale, 1.0)


def test_bias_act_kernel():
    from dfx import ops
    g = torch.Generator().manual_seed(2)
    for shape in ((2, 5, 7, 9), (3, 64, 50, 84), (1, 3, 1, 1)):
        x = torch.randn(*shape, generator=g).cuda()
        b = torch.randn(shape[1], generator=g).cuda()
        r = torch.randn(*shape, generator=g).cuda()
        for res in (None, r):
            for relu in (False, True):
                want = x + b.view(1, -1, 1, 1) + (0 if res is None else res)
                want = want.relu() if relu else want
                got = ops.bias_act_(x.clone(), b, res, relu)
                assert torch.allclose(got, want, atol=1e-6)


def _cpu_ops():
    import models.ops.functions.ms_deform_attn_func as f
    from dfx import ops
    from tests.test_clip_shard_gloo import _patch_cpu_ops
    saved = (f.MSDeformAttnFunction, ops.roi_align)
    _patch_cpu_ops()
    return f, ops, saved


@pytest.mark.parametrize("fusion", ["Baseline", "LateFusion", "Encoder_CrossFusion"])
def test_single_frame_configs_gpu_vs_cpu_oracle(fusion):
    """BASELINE.json configs A-C: Deformable-DETR single frame (RGB / Late Fusion / Encoder Cross Fusion)
    through build_model, padded batch of two different-size images (valid_ratio != 1 branches),
    HIP path vs the same host code on CPU with the oracle operator; box indices of PostProcess."""
    from models import build_model
    from models.config import single_args
    from models.fused import enable_fused_inference
    from tests._param_fill import fill_params_by_name
    from util.misc import nested_tensor_from_tensor_list
    C = 3 if fusion == "Baseline" else 4
    g = torch.Generator().manual_seed(9)
    imgs = [torch.randn(C, 64, 96, generator=g), torch.randn(C, 48, 80, generator=g)]

    def make(device):
        model, _, post = build_model(single_args(fusion, device=device))
        fill_params_by_name(model, seed=8)
        with torch.no_grad():
            for h in model.bbox_embed:
                h.layers[-1].weight.mul_(0.2)
        return model.to(device).eval(), post

    gm, post = make("cuda")
    enable_fused_inference(gm)
    with torch.no_grad():
        got = gm(nested_tensor_from_tensor_list([i.cuda() for i in imgs]))
    f, ops, saved = _cpu_ops()
    try:
        cm, _ = make("cpu")
        with torch.no_grad():
            want = cm(nested_tensor_from_tensor_list(imgs))
    finally:
        f.MSDeformAttnFunction, ops.roi_align = saved
    lg, bx = got["pred_logits"].cpu(), got["pred_boxes"].cpu()
    assert lg.shape == (2, 300, 3)
    assert (lg - want["pred_logits"]).abs().max() < 1e-3 and (bx - want["pred_boxes"]).abs().max() < 1e-3
    sizes = torch.tensor([[64, 96], [48, 80]])
    for rg, rc in zip(post["bbox"]({"pred_logits": lg, "pred_boxes": bx}, sizes), post["bbox"](want, sizes)):
        margin = (rc["scores"][:-1] - rc["scores"][1:]).abs()
        safe = torch.ones(100, dtype=torch.bool)
        safe[:-1] &= margin > 1e-5
        safe[1:] &= margin > 1e-5
        assert torch.equal(rg["labels"][safe], rc["labels"][safe])
        assert torch.allclose(rg["boxes"][safe], rc["boxes"][safe], atol=0.2)


def test_padded_clip_gpu_vs_cpu_oracle():
    """A TransVOD++ clip whose frames carry real padding (mask != 0, valid ratios < 1): the spatial stage
    (backbones, Late Fusion, encoder, decoder, heads) on the HIP path vs the CPU oracle path.

    Only quantities that do not read PADDED tokens are compared.  The sine positional embedding of a
    fully padded row / column is sin(~1e6) (normalisation by eps, position_encoding.py:47-49), whose
    value depends on the libm at hand; masked attention never looks at those tokens, but the
    query/RoI fusion of the temporal stage pools RoIs over the whole map including them, so its output
    on padded clips is device-dependent in the reference as well."""
    from models.clip_inference import ClipRunner
    clip = _clip(3, 33)
    mask = torch.zeros(3, 64, 96, dtype=torch.bool)
    mask[:, 56:, :] = True
    mask[:, :, 80:] = True
    clip = clip * (~mask)[:, None]
    gm, _ = _build("cuda")
    got = ClipRunner(gm.cuda(), micro_batch=3).frames_forward(clip.cuda(), mask.cuda())
    f, ops, saved = _cpu_ops()
    try:
        cm, _ = _build("cpu")
        want = ClipRunner(cm, micro_batch=3).frames_forward(clip, mask)
    finally:
        f.MSDeformAttnFunction, ops.roi_align = saved
    assert (got["logits"].cpu() - want["logits"]).abs().max() < 1e-3
    assert (got["ref_last"].cpu() - want["ref_last"]).abs().max() < 1e-3
    valid = ~torch.nn.functional.interpolate(mask[None].float(), size=(4, 6)).bool()[0].flatten(1)      # stride-16 map
    diff = (got["memory"].cpu() - want["memory"]).abs().max(-1)[0]
    assert (diff * valid).max() < 1e-3


def test_config_a_single_image_794x600_gpu_vs_cpu_oracle():
    """BASELINE.json configs[0]: single-frame RGB Deformable-DETR on ONE image at the size the reference's
    sample_dataset/OID image takes after its resize (773x1024 -> short side 600: [1,3,794,600], 1900 tokens after
    DC5): odd map sizes all the way down (H*W % 4 != 0 on some stages: the implicit-GEMM route of the 1x1
    convolutions, Winograd edge tiles).  HIP path vs the same host code on CPU with the oracle operators;
    PostProcess labels / box indices index by index outside a 2e-5 tie margin."""
    from models import build_model
    from models.config import single_args
    from models.fused import enable_fused_inference
    from tests._cases_detector import compare_indices
    from tests._param_fill import fill_params_by_name
    from util.misc import nested_tensor_from_tensor_list
    img = torch.randn(3, 794, 600, generator=torch.Generator().manual_seed(21))

    def make(device):
        model, _, post = build_model(single_args("Baseline", device=device))
        fill_params_by_name(model, seed=8)
        with torch.no_grad():
            for h in model.bbox_embed:
                h.layers[-1].weight.mul_(0.2)
        return model.to(device).eval(), post

    gm, post = make("cuda")
    enable_fused_inference(gm)
    with torch.no_grad():
        got = gm(nested_tensor_from_tensor_list([img.cuda()]))
    f, ops, saved = _cpu_ops()
    try:
        cm, _ = make("cpu")
        with torch.no_grad():
            want = cm(nested_tensor_from_tensor_list([img]))
    finally:
        f.MSDeformAttnFunction, ops.roi_align = saved
    lg, bx = got["pred_logits"].cpu(), got["pred_boxes"].cpu()
    assert lg.shape == (1, 300, 3)
    assert (lg - want["pred_logits"]).abs().max() < 1e-3 and (bx - want["pred_boxes"]).abs().max() < 1e-3
    sizes = torch.tensor([[794, 600]])
    rg, rc = post["bbox"]({"pred_logits": lg, "pred_boxes": bx}, sizes)[0], post["bbox"](want, sizes)[0]
    n, bad = compare_indices(rc["labels"][None], rg["labels"][None], rc["scores"][None], 2e-5)
    assert n > 10 and bad == 0
    idx_c = torch.topk(want["pred_logits"].sigmoid().flatten(1), 100, dim=1)[1] // 3
    idx_g = torch.topk(lg.sigmoid().flatten(1), 100, dim=1)[1] // 3
    n, bad = compare_indices(idx_c, idx_g, rc["scores"][None], 2e-5)
    assert n > 10 and bad == 0
