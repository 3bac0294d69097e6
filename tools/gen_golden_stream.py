"""Generate the VIDEO-inference fixture from the REFERENCE's own caller and detector (build container only):
    python tools/gen_golden_stream.py   ->   tests/golden/stream.npz
Per synthetic video of tests/_cases_stream.py and per frame: the reference's ``get_image_and_reference_clips``
(inference.py:721-794; reference-frame window, repetition, channel assembly) on a synthetic COCO-VID index whose images
are the video's frames, its ``util.misc_multi.nested_tensor_from_tensor_list`` (clip split), and the forward of its
TransVOD++ ``DeformableDETR`` (stub backbones of tests/_cases_stream.py; RoIAlign = this repository's oracle restatement on
both sides, as in every TransVOD++ fixture).  Stored: pred_logits / pred_boxes of every frame, and the ids of the
reference frames the reference picked."""
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import ref_import  # noqa: E402
from oracle import msda_oracle  # noqa: E402

ref_import.install(roi_align_fn=lambda x, rois, size, scale, ratio, aligned: msda_oracle.roi_align(x, rois, size, scale, ratio, aligned))
ref = ref_import.import_inference()
import models.deformable_detr_multi_plusplus as multipp  # noqa: E402
import models.deformable_transformer_multi_plusplus as tpp  # noqa: E402
import util.misc_multi as utils  # noqa: E402
from models.position_encoding import PositionEmbeddingSine  # noqa: E402

from tests._cases_stream import VIDEOS, build_detector, video_frames  # noqa: E402

ns = SimpleNamespace(multipp=multipp, tpp=tpp, NestedTensorMulti=utils.NestedTensor, PositionEmbeddingSine=PositionEmbeddingSine)
torch.set_grad_enabled(False)
blobs = {}
for name, v in VIDEOS.items():
    frames = video_frames(name)
    n, R, depth = v["n"], v["R"], v["depth"]
    det = build_detector(ns, R, depth)

    class FakeCoco:
        imgs = {i: None for i in range(1, n + 1)}        # image ids 1..n, one video

        def getAnnIds(self, imgIds):
            return []

        def loadAnns(self, ids):
            return []

        def loadImgs(self, i):
            return [{"file_name": f"{i}", "video_id": 1}]

    class FakeVid:
        def get_img_ids_from_vid(self, vid):
            return list(range(1, n + 1))

    me = object.__new__(ref.DeformableDETR)              # inference.py's caller class
    me.coco, me.cocovid = FakeCoco(), FakeVid()
    me.num_ref_frames, me.filter_key_img, me.depth_available, me.img_path = R, True, depth, ""
    me.get_image = lambda path: int(path)
    me.get_depth = lambda path: int(path)
    me.prepare = lambda img, d, target: (img, d, target)
    me.rgb_transform = lambda i: torch.cat([frames[i - 1][:3], torch.full((3, 1, frames.shape[3]), float(i))], 1)   # id rides in an extra row
    me.depth_transform = lambda i: torch.cat([frames[i - 1][3:4], torch.zeros(1, 1, frames.shape[3])], 1)
    logits, boxes, picked = [], [], []
    for idx in range(n):
        clip, _, _ = ref.DeformableDETR.get_image_and_reference_clips(me, idx)
        C = 4 if depth else 3
        picked.append(clip[0::C, -1, 0].long() - 1)                                       # frame ids of the clip, in order
        clip = clip[:, :-1]                                                              # drop the id row
        out = det(utils.nested_tensor_from_tensor_list([clip], channel_size=C))
        logits.append(out["pred_logits"][0])
        boxes.append(out["pred_boxes"][0])
    blobs[f"{name}.pred_logits"] = torch.stack(logits).numpy()
    blobs[f"{name}.pred_boxes"] = torch.stack(boxes).numpy()
    blobs[f"{name}.clip_frame_ids"] = torch.stack(picked).numpy()
    print(name, torch.stack(picked).tolist())
OUT = os.path.join(ROOT, "tests", "golden", "stream.npz")
np.savez_compressed(OUT, **blobs)
print("wrote", OUT, f"{os.path.getsize(OUT) / 1e3:.1f} kB")
