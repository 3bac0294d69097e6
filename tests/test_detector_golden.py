"""This repository's DETECTOR classes (DeformableDETR single-frame / TransVOD++, PostProcess, the inference filter)
against outputs of the REFERENCE's classes on the same stub-backbone features, seeded inputs and name-keyed weights
(tests/_cases_detector.py, tools/gen_golden_detector.py -> tests/golden/detector.npz).

Floating outputs within 2e-4 (north star: 1e-3); int64 outputs - PostProcess labels and box indices, the ordered
temporal top-k picks - torch.equal wherever the reference's scores at that rank are separated from their neighbours
by more than TIE_MARGIN (a near-tie may swap under any change of fp32 summation order, on the reference's own
hardware as well); the keep mask equal wherever the probability is further than TIE_MARGIN from the threshold.
CPU run: the oracle as the MSDA / RoIAlign operator.  GPU run (-m gpu): the HIP kernels, fused inference routes.
"""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from tests._cases_detector import compare_indices, run_detector_cases

TIE_MARGIN = 2e-5


def my_namespace():
    import models.deformable_detr_multi as multi
    import models.deformable_detr_multi_plusplus as multipp
    import models.backbone_scratch as bsc
    import models.deformable_detr_single as single
    import models.deformable_transformer_multi as tm
    import models.deformable_transformer_multi_plusplus as tpp
    import models.deformable_transformer_single as ts
    from models.position_encoding import PositionEmbeddingSine
    from util.misc import NestedTensor
    from util.misc_multi import NestedTensor as NestedTensorMulti
    return SimpleNamespace(bsc=bsc, single=single, multipp=multipp, multi=multi, ts=ts, tpp=tpp, tm=tm, NestedTensor=NestedTensor,
                           NestedTensorMulti=NestedTensorMulti, PositionEmbeddingSine=PositionEmbeddingSine)


@pytest.fixture(scope="module")
def golden(golden_dir):
    return np.load(os.path.join(golden_dir, "detector.npz"))


def check(got, golden):
    assert set(got) == set(golden.files)
    report = {}
    for key in sorted(got):
        ref, out = torch.from_numpy(golden[key]), got[key]
        if key.endswith(".state_dict_json"):
            # checkpoint wire format: the reference detector's state_dict keys and shapes, exactly
            import json
            want, have = json.loads(bytes(ref.tolist()).decode()), json.loads(bytes(out.tolist()).decode())
            assert len(want) > 20
            missing, extra = sorted(set(want) - set(have)), sorted(set(have) - set(want))
            assert not missing and not extra, f"{key}: missing {missing[:5]} extra {extra[:5]}"
            wrong = [k for k in want if want[k] != have[k]]
            assert not wrong, f"{key}: shapes differ for {wrong[:5]}"
            continue
        assert out.shape == ref.shape and out.dtype == ref.dtype, key
        if ref.dtype == torch.bool and ".mask" in key:
            assert torch.equal(out, ref), key
        if ref.is_floating_point():
            diff = (out - ref).abs()
            if ".pos" in key:
                # sine positions of fully padded rows / columns are sin / cos of ~ -3e6 (0.5 / eps): ill-conditioned in
                # fp32 and never read (masked); compared on the valid pixels
                valid = ~torch.from_numpy(golden[key.replace(".pos", ".mask")])
                diff = diff * valid[:, None].to(diff.dtype)
            err = diff.max().item()
            scale = 1.0 if "pp_boxes" not in key else 640.0          # PostProcess boxes are in pixels
            assert err < 2e-4 * scale, f"{key}: max abs err {err:.3e}"
            report[key] = err
    # int64: ordered index tensors, compared outside the tie margin
    for case in ("det_single", "det_multipp", "det_multipp_rgb", "det_multi", "det_multi_rgb"):
        scores = torch.from_numpy(golden[f"{case}.pp_scores"])
        for name in ("pp_labels", "pp_box_idx"):
            n, bad = compare_indices(torch.from_numpy(golden[f"{case}.{name}"]), got[f"{case}.{name}"], scores, TIE_MARGIN)
            # the name-keyed weights spread the scores: at least 80 % of the ranks are compared index by index
            assert n >= 0.8 * scores.numel() and bad == 0, f"{case}.{name}: {bad} of {n} clear ranks differ"
            report[f"{case}.{name}.ranks_compared"] = n / scores.numel()
        p = torch.from_numpy(golden[f"{case}.keep_probas"])
        clear = (p - float(golden[f"{case}.keep_prob"])).abs() > TIE_MARGIN
        assert torch.equal(got[f"{case}.keep_mask"][clear], torch.from_numpy(golden[f"{case}.keep_mask"])[clear])
        assert 0 < int(golden[f"{case}.keep_mask"].sum()) < p.numel(), "the fixture's keep mask must be non-trivial"
    for case, i in ((c, i) for c in ("det_multipp", "det_multipp_rgb", "det_multi", "det_multi_rgb") for i in range(3)):
        ref_idx, vals = torch.from_numpy(golden[f"{case}.topk{i}_idx"]), torch.from_numpy(golden[f"{case}.topk{i}_values"])
        n, bad = compare_indices(ref_idx, got[f"{case}.topk{i}_idx"], vals, TIE_MARGIN)
        assert n >= 0.8 * ref_idx.numel() and bad == 0, f"{case} temporal top-k {i}: {bad} of {n} clear ranks differ"
        report[f"{case}.topk{i}.ranks_compared"] = n / ref_idx.numel()
        # as a set the pick may only differ in candidates within the margin of the cut
        cut = vals[0, -1].item()
        a, b = set(ref_idx[0].tolist()), set(got[f"{case}.topk{i}_idx"][0].tolist())
        gvals = dict(zip(got[f"{case}.topk{i}_idx"][0].tolist(), got[f"{case}.topk{i}_values"][0].tolist()))
        rvals = dict(zip(ref_idx[0].tolist(), vals[0].tolist()))
        for j in a ^ b:
            assert abs((rvals.get(j) if j in rvals else gvals[j]) - cut) <= TIE_MARGIN, f"{case} top-k {i}: index {j} is not a tie at the cut"
    return report


def test_detectors_match_the_reference_cpu(golden, cpu_msda, monkeypatch, oracle):
    from dfx import ops

    def roi_align(inp, rois, output_size, spatial_scale, sampling_ratio, aligned=True, channels_last=False):
        size = output_size if isinstance(output_size, int) else output_size[0]
        if channels_last:
            out = oracle.roi_align(inp.permute(0, 3, 1, 2).contiguous(), rois, size, spatial_scale, sampling_ratio, aligned)
            return out.flatten(2).transpose(1, 2).contiguous()
        return oracle.roi_align(inp, rois, size, spatial_scale, sampling_ratio, aligned)

    monkeypatch.setattr(ops, "roi_align", roi_align)
    with torch.no_grad():
        got = run_detector_cases(my_namespace())
    print({k: f"{v:.1e}" for k, v in check(got, golden).items()})


@pytest.mark.gpu
def test_detectors_match_the_reference_gpu(golden):
    from models.fused import enable_fused_inference
    import tests._cases_detector as cd
    orig = cd.fill_params_by_name

    def fill_and_fuse(module, seed=0, prefix=""):
        enable_fused_inference(module, True)      # fused routes of every sub-module that has one
        return orig(module, seed=seed, prefix=prefix)

    cd.fill_params_by_name = fill_and_fuse
    try:
        with torch.no_grad():
            got = run_detector_cases(my_namespace(), device="cuda")
    finally:
        cd.fill_params_by_name = orig
    print({k: f"{v:.1e}" for k, v in check(got, golden).items()})
