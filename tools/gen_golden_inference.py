"""Generate golden vectors for the inference CALLER's helper steps from the REFERENCE's inference.py
(build container only):   python tools/gen_golden_inference.py   ->  tests/golden/inference_io.npz

inference.py imports cv2, pycocotools, matplotlib, tqdm and torchvision.transforms at module level; none exists in this
image and none is used by the helpers pinned here, so they are satisfied by EMPTY placeholder modules (the recipe of
SURVEY.md 8c / tools/ref_import.py extended by import-only names).  What is called is the reference's own code:

  * ``resize`` -> ``get_size_with_aspect_ratio``                         inference.py:310-353  (size rule: short side / max side)
  * ``DeformableDETR.rescale_bboxes`` / ``box_cxcywh_to_xyxy``           inference.py:456-489
  * ``DeformableDETR.get_image_and_reference_clips``                     inference.py:721-794  (reference-frame window,
    repetition, key-frame filter, [ (1+R)*C, H, W ] channel assembly) on a synthetic COCO-VID index whose "images"
    are constant tensors carrying their image id, so the output encodes which frames were sampled in which order.
  * ``util.misc`` / ``util.misc_multi``: ``nested_tensor_from_tensor_list`` and ``collate_fn`` on ragged images and
    clips (zero padding, padding masks, the channel split of a [(1+R)*C, H, W] clip into frames).   util/misc.py:304-356,
                                                                                                    util/misc_multi.py:304-345
  * the per-image tail of ``DeformableDETR.infer``: ``softmax(-1)[0][:, 1] > keep_prob`` filter and the label file
    ``Hand cx cy w h p`` (8 decimals) it writes - inline code, executed from the reference's file on synthetic model
    outputs.                                                                                      inference.py:918-956
  * the checkpoint resume / merge block of main_multi.py (``if args.resume:`` ..., eval branch: temporal keys taken from
    the TransVOD checkpoint per dataset type, spatial checkpoint laid over, ``strict=False`` load, thop counters
    filtered from the report) - inline code of ``main()``, executed from the reference's file at generation time on
    the synthetic checkpoints of tests/_cases_checkpoint.py.                                       main_multi.py:332-381
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import ref_import  # noqa: E402

ref_import.install(roi_align_fn=None)
ref = ref_import.import_inference()


blobs = {}

# ---- size rule -------------------------------------------------------------------------------------------------------
SIZE_CASES = [(1024, 773, 600, 1333), (773, 1024, 600, 1333), (1333, 800, 800, 1333), (1920, 1080, 800, 1333), (640, 480, 600, 1333),
              (500, 500, 600, 1333), (3000, 400, 600, 1333), (400, 3000, 600, 1333), (600, 900, 600, 1333), (999, 562, 600, 1000)]
out = []
for w, h, size, max_size in SIZE_CASES:
    tag, (oh, ow) = ref.resize(types.SimpleNamespace(size=(w, h)), size, max_size)
    out.append((w, h, size, max_size, oh, ow))
blobs["resize.cases"] = np.asarray(out, dtype=np.int64)

# ---- box rescale -----------------------------------------------------------------------------------------------------
me = object.__new__(ref.DeformableDETR)
boxes = torch.rand(37, 4, generator=torch.Generator().manual_seed(5))
blobs["rescale.boxes"] = boxes.numpy()
blobs["rescale.xyxy_640x480"] = ref.DeformableDETR.rescale_bboxes(me, boxes, (640, 480)).numpy()
blobs["rescale.xyxy_1333x800"] = ref.DeformableDETR.rescale_bboxes(me, boxes, (1333, 800)).numpy()

# ---- reference-frame sampling + clip assembly --------------------------------------------------------------------------
VIDEOS = {1: list(range(1, 8)), 2: list(range(8, 10)), 3: list(range(10, 31))}      # image ids per video


class FakeCoco:
    imgs = {i: None for ids in VIDEOS.values() for i in ids}

    def getAnnIds(self, imgIds):
        return []

    def loadAnns(self, ids):
        return []

    def loadImgs(self, i):
        vid = next(v for v, ids in VIDEOS.items() if i in ids)
        return [{"file_name": f"{i}", "video_id": vid}]


class FakeVid:
    def get_img_ids_from_vid(self, vid):
        return VIDEOS[vid]


def sampled(num_ref, filter_key, depth):
    me = object.__new__(ref.DeformableDETR)
    me.coco, me.cocovid = FakeCoco(), FakeVid()
    me.num_ref_frames, me.filter_key_img, me.depth_available, me.img_path = num_ref, filter_key, depth, ""
    me.get_image = lambda path: float(path)
    me.get_depth = lambda path: -float(path)
    me.prepare = lambda img, d, target: (img, d, target)
    me.rgb_transform = lambda v: torch.full((3, 2, 2), v)
    me.depth_transform = lambda v: torch.full((1, 2, 2), v)
    rows = []
    ids = sorted(FakeCoco.imgs)
    for idx in range(len(ids)):
        clip, target, path = ref.DeformableDETR.get_image_and_reference_clips(me, idx)
        rows.append(clip[:, 0, 0])
    return torch.stack(rows)


for num_ref, filter_key, depth in ((4, True, True), (2, False, False), (31, True, True), (1, True, False)):
    blobs[f"clips.R{num_ref}_filter{int(filter_key)}_depth{int(depth)}"] = sampled(num_ref, filter_key, depth).numpy()
blobs["clips.video_of_image"] = np.asarray([[i, v] for v, ids in VIDEOS.items() for i in ids], dtype=np.int64)

# ---- checkpoint merge (row f3): the reference's own resume block of main_multi.py, eval branch ---------------------------
# The block is inline code of main() (not importable), so its source lines are read from the reference at generation time
# and executed here on synthetic checkpoints (tests/_cases_checkpoint.py); nothing of it is stored in the repository.
import json  # noqa: E402
import tempfile  # noqa: E402
import textwrap  # noqa: E402

from tests._cases_checkpoint import CASES, Target, describe, write_checkpoints  # noqa: E402

src = open(os.path.join(ref_import.REF, "main_multi.py")).read().splitlines()
start = next(i for i, ln in enumerate(src) if ln.strip() == "if args.resume:")
stop = next(i for i, ln in enumerate(src) if i > start and ln.strip().startswith("unexpected_keys = [k for k in unexpected_keys"))
block = compile(textwrap.dedent("\n".join(src[start:stop + 1])), "main_multi.py resume block", "exec")
results = []
with tempfile.TemporaryDirectory() as tmp:
    paths = write_checkpoints(tmp)
    for dataset_file, with_temporal, with_spatial in CASES:
        target = Target()
        args = types.SimpleNamespace(resume=paths["base"], dataset_file=dataset_file, eval=True, coco_pretrain=False,
                                     transvod_temporal_weights=paths["temporal"] if with_temporal else None,
                                     spatial_weights=paths["spatial"] if with_spatial else None)
        scope = {"torch": torch, "args": args, "model_without_ddp": target, "print": lambda *a, **k: None}
        exec(block, scope)
        results.append(describe(target, scope["missing_keys"], scope["unexpected_keys"]))
blobs["checkpoint.merge_json"] = np.frombuffer(json.dumps(results, sort_keys=True).encode(), dtype=np.uint8)

# ---- collate (row a17): ragged images -> padded batch + mask, single-frame and clip (channel-split) forms -----------------
import util.misc as ref_misc  # noqa: E402
import util.misc_multi as ref_misc_multi  # noqa: E402

from tests._cases_checkpoint import collate_inputs  # noqa: E402

imgs, clips = collate_inputs()
nt = ref_misc.nested_tensor_from_tensor_list(imgs)
blobs["collate.single_tensors"], blobs["collate.single_mask"] = nt.tensors.numpy(), nt.mask.numpy()
batch = ref_misc.collate_fn([(im, {"i": i}) for i, im in enumerate(imgs)])
blobs["collate.single_collate_fn_tensors"] = batch[0].tensors.numpy()
for split, cs, tag in ((True, 4, "rgbd"), (True, 3, "rgb"), (False, 3, "nosplit")):
    nt = ref_misc_multi.nested_tensor_from_tensor_list(clips[tag], split=split, channel_size=cs)
    blobs[f"collate.multi_{tag}_tensors"], blobs[f"collate.multi_{tag}_mask"] = nt.tensors.numpy(), nt.mask.numpy()
batch = ref_misc_multi.collate_fn([(c, {"i": i}) for i, c in enumerate(clips["rgbd"])], use_depth=True)
blobs["collate.multi_collate_fn_tensors"], blobs["collate.multi_collate_fn_mask"] = batch[0].tensors.numpy(), batch[0].mask.numpy()

# ---- box conversions (row a17) ------------------------------------------------------------------------------------------
import util.box_ops as ref_box  # noqa: E402

bx = torch.rand(29, 4, generator=torch.Generator().manual_seed(9))
blobs["boxops.cxcywh"] = bx.numpy()
blobs["boxops.to_xyxy"] = ref_box.box_cxcywh_to_xyxy(bx).numpy()
blobs["boxops.back_to_cxcywh"] = ref_box.box_xyxy_to_cxcywh(ref_box.box_cxcywh_to_xyxy(bx)).numpy()

# ---- post-filter + label file (row f1): the per-image tail of the reference's infer() loop --------------------------------
# inline code of DeformableDETR.infer (inference.py, "probas = model_outputs['pred_logits']..." to the label f.write): read
# from the reference at generation time and executed on synthetic model outputs; the label files it writes are the fixture.
from pathlib import Path  # noqa: E402

isrc = open(os.path.join(ref_import.REF, "inference.py")).read().splitlines()
i0 = next(i for i, ln in enumerate(isrc) if ln.strip().startswith("probas = model_outputs['pred_logits'].softmax(-1)[0]"))
i1 = next(i for i, ln in enumerate(isrc) if i > i0 and "f.write(f'Hand " in ln)
tail = compile("for _once in (0,):\n" + textwrap.indent(textwrap.dedent("\n".join(isrc[i0:i1 + 1])), "    "),
               "inference.py infer() tail", "exec")
label_cases = []
with tempfile.TemporaryDirectory() as tmp:
    for case, (seed, keep_prob) in enumerate(((1, 0.5), (2, 0.3), (3, 0.999), (4, 0.7))):
        g = torch.Generator().manual_seed(seed)
        outputs = {"pred_logits": torch.randn(1, 40, 3, generator=g) * 2.0, "pred_boxes": torch.rand(1, 40, 4, generator=g)}
        me = object.__new__(ref.DeformableDETR)
        me.args = types.SimpleNamespace(keep_prob=keep_prob)
        me.output_dir, me.img_path, me.save_txt, me.save_fig, me.depth_available = os.path.join(tmp, str(case)), tmp, True, False, False
        me.plot_results = lambda *a, **k: None
        scope = {"torch": torch, "os": os, "Path": Path, "tqdm": types.SimpleNamespace(write=lambda *a, **k: None), "self": me,
                 "model_outputs": outputs, "original_img": types.SimpleNamespace(size=(640, 480)), "original_dpth": None,
                 "img_file": os.path.join(tmp, "vid", f"frame{case}.jpg")}
        exec(tail, scope)
        label = os.path.join(me.output_dir, "labels", "vid", f"frame{case}.txt")
        label_cases.append({"seed": seed, "keep_prob": keep_prob,
                            "lines": open(label).read().splitlines() if os.path.exists(label) else None})
assert any(c["lines"] is None for c in label_cases) and any(c["lines"] for c in label_cases)
blobs["labels.cases_json"] = np.frombuffer(json.dumps(label_cases).encode(), dtype=np.uint8)

# ---- DFormer partial load (row f3): the reference's DFormerBackbone.load_pretrained_weights ------------------------------
import models.dformer_backbone as ref_dfb  # noqa: E402

from tests._cases_checkpoint import describe_dformer, write_dformer_checkpoint  # noqa: E402

with tempfile.TemporaryDirectory() as tmp:
    back = ref_dfb.build_dformer_backbone(types.SimpleNamespace(hidden_dim=256, position_embedding="sine", dformer_weights=None))[0]
    for prm in back.depth_backbone.state_dict().values():
        prm.zero_()
    path = os.path.join(tmp, "dformer.pth")
    write_dformer_checkpoint(path, back.depth_backbone)
    back.load_pretrained_weights(back.depth_backbone, path)
    blobs["checkpoint.dformer_json"] = np.frombuffer(json.dumps(describe_dformer(back.depth_backbone), sort_keys=True).encode(), dtype=np.uint8)

OUT = os.path.join(ROOT, "tests", "golden", "inference_io.npz")
np.savez_compressed(OUT, **blobs)
print("wrote", OUT, {k: v.shape for k, v in blobs.items()})
