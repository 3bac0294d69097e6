"""CPU: the caller-side steps around the path."""
import torch

from models import inference_io as io


def test_reference_frame_sampling_rule():
    ids = list(range(100, 120))
    assert io.sample_reference_ids(110, ids, 4) == [106, 107, 108, 109]            # window minus the key frame
    assert io.sample_reference_ids(100, ids, 4) == [101, 102, 103, 104]            # clipped at the start
    assert io.sample_reference_ids(119, ids, 4) == [115, 116, 117, 118]
    assert io.sample_reference_ids(110, ids, 4, filter_key_img=False) == [106, 107, 108, 109]
    assert io.sample_reference_ids(5, [4, 5], 3) == [4, 4, 4]                       # short video: repeat


def test_clip_assembly_round_trips_through_the_collate():
    from util.misc_multi import nested_tensor_from_tensor_list
    g = torch.Generator().manual_seed(0)
    rgb = [torch.randn(3, 8, 10, generator=g) for _ in range(3)]
    dep = [torch.randn(1, 8, 10, generator=g) for _ in range(3)]
    clip = io.assemble_clip(rgb, dep)
    assert clip.shape == (12, 8, 10)
    nt = nested_tensor_from_tensor_list([clip], split=True, channel_size=4)
    assert nt.tensors.shape == (3, 4, 8, 10)
    for t in range(3):
        assert torch.equal(nt.tensors[t, :3], rgb[t]) and torch.equal(nt.tensors[t, 3:], dep[t])
    assert io.assemble_clip(rgb).shape == (9, 8, 10)


def test_post_filter_rescale_and_label_lines():
    logits = torch.tensor([[[0.0, 3.0, 0.0], [2.0, 0.0, 0.0], [0.0, 0.2, 0.0]]])
    boxes = torch.tensor([[[0.5, 0.5, 0.2, 0.4], [0.1, 0.1, 0.1, 0.1], [0.3, 0.6, 0.2, 0.2]]])
    probs, kept, idx = io.filter_detections({"pred_logits": logits, "pred_boxes": boxes}, keep_prob=0.5)
    assert idx.tolist() == [0] and torch.allclose(probs, logits.softmax(-1)[0, :1, 1])
    px = io.rescale_bboxes(kept, (200, 100))
    assert torch.allclose(px, torch.tensor([[80.0, 30.0, 120.0, 70.0]]))
    line = io.yolo_lines(kept, probs)[0].split()
    assert line[0] == "Hand" and len(line) == 6 and abs(float(line[1]) - 0.5) < 1e-7


def test_checkpoint_merge_and_load(tmp_path):
    """{'model': state_dict} wire format; temporal modules come from the TransVOD++ checkpoint, the
    spatial fine-tune overlays everything it holds; keys are the reference's."""
    torch.manual_seed(0)

    class Tiny(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.backbone = torch.nn.Linear(2, 2)
            self.temporal_query_layer1 = torch.nn.Linear(2, 2)
            self.dynamic_layer_for_current_query1 = torch.nn.Linear(2, 2)
            self.temp_bbox_embed = torch.nn.Linear(2, 2)
            self.class_embed = torch.nn.Linear(2, 2)

    base, temporal, spatial, target = Tiny(), Tiny(), Tiny(), Tiny()
    p = {n: str(tmp_path / f"{n}.pth") for n in ("base", "temporal", "spatial")}
    torch.save({"model": base.state_dict(), "epoch": 3}, p["base"])
    torch.save({"model": {**temporal.state_dict(), "x.total_ops": torch.zeros(1)}}, p["temporal"])
    torch.save({"model": {k: v for k, v in spatial.state_dict().items() if k.startswith("backbone")}}, p["spatial"])
    missing, unexpected = io.load_checkpoint(target, p["base"], p["spatial"], p["temporal"], "vid_multi_plusplus")
    assert missing == [] and unexpected == []
    sd = target.state_dict()
    assert torch.equal(sd["backbone.weight"], spatial.state_dict()["backbone.weight"])
    assert torch.equal(sd["temporal_query_layer1.weight"], temporal.state_dict()["temporal_query_layer1.weight"])
    assert torch.equal(sd["dynamic_layer_for_current_query1.bias"], temporal.state_dict()["dynamic_layer_for_current_query1.bias"])
    assert torch.equal(sd["temp_bbox_embed.weight"], temporal.state_dict()["temp_bbox_embed.weight"])
    assert torch.equal(sd["class_embed.weight"], base.state_dict()["class_embed.weight"])
    # TransVOD (vid_multi) does not move the dynamic_layer weights
    io.load_checkpoint(target, p["base"], None, p["temporal"], "vid_multi")
    assert torch.equal(target.state_dict()["dynamic_layer_for_current_query1.bias"],
                       base.state_dict()["dynamic_layer_for_current_query1.bias"])


def test_real_model_state_dict_round_trip(tmp_path):
    from models import build_model
    from models.config import single_args
    model, _, _ = build_model(single_args("LateFusion", device="cpu"))
    path = str(tmp_path / "ck.pth")
    torch.save({"model": model.state_dict()}, path)
    other, _, _ = build_model(single_args("LateFusion", device="cpu"))
    missing, unexpected = io.load_checkpoint(other, path)
    assert missing == [] and unexpected == []
    for (k, a), (_, b) in zip(model.state_dict().items(), other.state_dict().items()):
        assert torch.equal(a, b), k


def test_memo_on_tensor_identity_and_version():
    """util/memo.py: a value derived from a (mask) tensor is rebuilt when the tensor object or its version
    counter changes, shared otherwise, and never cached while autograd records."""
    import torch
    from util import memo
    memo.clear()
    calls = []

    def build(t):
        calls.append(1)
        return (~t).cumsum(1, dtype=torch.float32)

    m = torch.zeros(2, 5, dtype=torch.bool)
    with torch.no_grad():
        a = memo.memo_on(m, "x", lambda: build(m))
        b = memo.memo_on(m, "x", lambda: build(m))
        assert a is b and len(calls) == 1
        assert memo.memo_on(m, "y", lambda: build(m)) is not a and len(calls) == 2      # another tag
        m[0, 0] = True                                                                     # in-place edit: version bump
        c = memo.memo_on(m, "x", lambda: build(m))
        assert c is not a and len(calls) == 3 and c[0, 0] == 0
        m2 = m.clone()                                                                     # equal content, another object
        assert memo.memo_on(m2, "x", lambda: build(m2)) is not c and len(calls) == 4
        view = m[:1]
        view[0, 1] = True                                                                  # edit through a view bumps the base
        assert memo.memo_on(m, "x", lambda: build(m)) is not c and len(calls) == 5
    with torch.enable_grad():
        memo.memo_on(m, "x", lambda: build(m))
        assert len(calls) == 6                                                             # no caching under autograd
    memo.clear()


# ---- pinned against the REFERENCE's inference.py (tools/gen_golden_inference.py -> tests/golden/inference_io.npz) ----
def _golden():
    import os
    import numpy as np
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "inference_io.npz"))


def test_size_rule_matches_reference_resize():
    """inference.py:310-353 (short side `size`, long side capped at `max_size`, int truncation)."""
    from models.preprocess import get_size_with_aspect_ratio
    for w, h, size, max_size, oh, ow in _golden()["resize.cases"].tolist():
        assert tuple(get_size_with_aspect_ratio((w, h), size, max_size)) == (oh, ow)


def test_rescale_bboxes_matches_reference():
    """inference.py:456-489."""
    import torch
    from models.inference_io import rescale_bboxes
    g = _golden()
    boxes = torch.from_numpy(g["rescale.boxes"])
    for key, size in (("rescale.xyxy_640x480", (640, 480)), ("rescale.xyxy_1333x800", (1333, 800))):
        assert torch.equal(rescale_bboxes(boxes, size), torch.from_numpy(g[key]))


def test_reference_frame_sampling_and_clip_assembly_match_reference():
    """inference.py:721-794 run on a synthetic COCO-VID index whose images are constants carrying their id: the
    reference's output row for image i is [id, id, id, (-id)] for the key frame followed by its reference frames."""
    import torch
    from models.inference_io import assemble_clip, sample_reference_ids
    g = _golden()
    video_of = {int(i): int(v) for i, v in g["clips.video_of_image"]}
    videos = {}
    for i, v in video_of.items():
        videos.setdefault(v, []).append(i)
    for key in [k for k in g.files if k.startswith("clips.R")]:
        num_ref = int(key.split("R")[1].split("_")[0])
        filt, depth = "filter1" in key, "depth1" in key
        want = torch.from_numpy(g[key])
        for row, img_id in enumerate(sorted(video_of)):
            refs = sample_reference_ids(img_id, sorted(videos[video_of[img_id]]), num_ref, filter_key_img=filt)
            ids = [img_id] + refs
            rgb = [torch.full((3, 2, 2), float(i)) for i in ids]
            dep = [torch.full((1, 2, 2), -float(i)) for i in ids] if depth else None
            assert torch.equal(assemble_clip(rgb, dep)[:, 0, 0], want[row]), (key, img_id)


def test_checkpoint_merge_matches_the_reference_resume_block(tmp_path):
    """Row f3: the same synthetic checkpoints through models.inference_io.load_checkpoint and through the reference's own
    resume block (main_multi.py:332-381, executed at fixture generation): every tensor of the model must come from the
    same file (1 = resumed checkpoint, 2 = TransVOD temporal checkpoint, 3 = spatial fine-tune, 0 = left untouched) and
    the missing / unexpected key reports must agree, for every dataset type and combination of companion files."""
    import json

    from tests._cases_checkpoint import CASES, Target, describe, write_checkpoints
    want = json.loads(bytes(_golden()["checkpoint.merge_json"].tolist()).decode())
    assert len(want) == len(CASES)
    paths = write_checkpoints(str(tmp_path))
    for (dataset_file, with_temporal, with_spatial), ref in zip(CASES, want):
        target = Target()
        missing, unexpected = io.load_checkpoint(target, paths["base"], paths["spatial"] if with_spatial else None,
                                                 paths["temporal"] if with_temporal else None, dataset_file)
        got = describe(target, missing, unexpected)
        assert got == ref, (dataset_file, with_temporal, with_spatial,
                            {k: (got["origin"][k], ref["origin"][k]) for k in ref["origin"] if got["origin"][k] != ref["origin"][k]},
                            got["missing"], ref["missing"], got["unexpected"], ref["unexpected"])
    # the fixture exercises every rule: temporal keys moved, dynamic_layer only for TransVOD++, spatial overlay wins
    assert {v for c in want for v in c["origin"].values()} == {0.0, 1.0, 2.0, 3.0}


def test_dformer_partial_load_matches_the_reference(tmp_path):
    """Row f3: DFormerBackbone.load_pretrained_weights on a synthetic DFormer checkpoint leaves every tensor of the depth
    stem with the value the reference's method leaves it with (depth-branch keys only, running statistics untouched, the
    shape rule for weights and the name rule for biases as they are in dformer_backbone.py:161-198)."""
    import json
    from types import SimpleNamespace

    from models.dformer_backbone import build_dformer_backbone
    from tests._cases_checkpoint import describe_dformer, write_dformer_checkpoint
    want = json.loads(bytes(_golden()["checkpoint.dformer_json"].tolist()).decode())
    back = build_dformer_backbone(SimpleNamespace(hidden_dim=256, position_embedding="sine", dformer_weights=None))[0]
    with torch.no_grad():
        for prm in back.depth_backbone.state_dict().values():
            prm.zero_()
    path = str(tmp_path / "dformer.pth")
    write_dformer_checkpoint(path, back.depth_backbone)
    back.load_pretrained_weights(back.depth_backbone, path)
    got = describe_dformer(back.depth_backbone)
    assert got == want, {k: (got.get(k), want.get(k)) for k in set(got) | set(want) if got.get(k) != want.get(k)}
    assert any(v > 0 for v in want.values()) and any(v == 0 for k, v in want.items() if "running" in k)


def test_label_files_match_the_reference_infer_tail():
    """Row f1: the kept detections and the label lines (`Hand cx cy w h p`, 8 decimals) equal, character by character, what
    the tail of the reference's infer() loop wrote for the same synthetic model outputs (inference.py:918-956, executed at
    fixture generation); an image with no detection above keep_prob gets no label file there and no lines here."""
    import json
    cases = json.loads(bytes(_golden()["labels.cases_json"].tolist()).decode())
    assert len(cases) == 4
    for c in cases:
        g = torch.Generator().manual_seed(c["seed"])
        outputs = {"pred_logits": torch.randn(1, 40, 3, generator=g) * 2.0, "pred_boxes": torch.rand(1, 40, 4, generator=g)}
        probs, boxes, idx = io.filter_detections(outputs, keep_prob=c["keep_prob"])
        lines = io.yolo_lines(boxes, probs)
        assert lines == (c["lines"] or []), (c["seed"], lines[:2], (c["lines"] or [])[:2])


def test_collate_matches_the_reference():
    """Row a17: ragged images / clips -> zero-padded batch + padding mask, bit for bit what the reference's util.misc and
    util.misc_multi build (single-frame form; clip form with the channel split into frames for RGB-D and RGB; no split)."""
    import util.misc as misc
    import util.misc_multi as misc_multi
    from tests._cases_checkpoint import collate_inputs
    g = _golden()
    imgs, clips = collate_inputs()
    nt = misc.nested_tensor_from_tensor_list(imgs)
    assert torch.equal(nt.tensors, torch.from_numpy(g["collate.single_tensors"])) and torch.equal(nt.mask, torch.from_numpy(g["collate.single_mask"]))
    batch = misc.collate_fn([(im, {"i": i}) for i, im in enumerate(imgs)])
    assert torch.equal(batch[0].tensors, torch.from_numpy(g["collate.single_collate_fn_tensors"])) and batch[1] == ({"i": 0}, {"i": 1}, {"i": 2})
    for split, cs, tag in ((True, 4, "rgbd"), (True, 3, "rgb"), (False, 3, "nosplit")):
        nt = misc_multi.nested_tensor_from_tensor_list(clips[tag], split=split, channel_size=cs)
        assert torch.equal(nt.tensors, torch.from_numpy(g[f"collate.multi_{tag}_tensors"])), tag
        assert torch.equal(nt.mask, torch.from_numpy(g[f"collate.multi_{tag}_mask"])), tag
    batch = misc_multi.collate_fn([(c, {"i": i}) for i, c in enumerate(clips["rgbd"])], use_depth=True)
    assert torch.equal(batch[0].tensors, torch.from_numpy(g["collate.multi_collate_fn_tensors"]))
    assert torch.equal(batch[0].mask, torch.from_numpy(g["collate.multi_collate_fn_mask"]))


def test_box_conversions_match_the_reference():
    from util import box_ops
    g = _golden()
    bx = torch.from_numpy(g["boxops.cxcywh"])
    assert torch.equal(box_ops.box_cxcywh_to_xyxy(bx), torch.from_numpy(g["boxops.to_xyxy"]))
    assert torch.equal(box_ops.box_xyxy_to_cxcywh(box_ops.box_cxcywh_to_xyxy(bx)), torch.from_numpy(g["boxops.back_to_cxcywh"]))
