"""Generate DETECTOR-level golden vectors from the REFERENCE's own classes (build container only):
    python tools/gen_golden_detector.py
Runs tests/_cases_detector.py against /root/reference's DeformableDETR (single-frame, TransVOD++ and TransVOD), PostProcess
and the inference filter rule, around stub backbones (the real ones need torchvision / downloads), and writes
tests/golden/detector.npz.  The MSDA operator inside the reference modules is the reference's
ms_deform_attn_core_pytorch behind the CUDA launcher's flat re-indexing (tools/ref_import.py); RoIAlign
(third-party mmcv, unpinned) is this repository's oracle restatement on both sides."""
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
import models.backbone_scratch as bsc  # noqa: E402
import models.deformable_detr_multi as multi  # noqa: E402
import models.deformable_detr_multi_plusplus as multipp  # noqa: E402
import models.deformable_detr_single as single  # noqa: E402
import models.deformable_transformer_multi as tm  # noqa: E402
import models.deformable_transformer_multi_plusplus as tpp  # noqa: E402
import models.deformable_transformer_single as ts  # noqa: E402
from models.position_encoding import PositionEmbeddingSine  # noqa: E402
from util.misc import NestedTensor  # noqa: E402
from util.misc_multi import NestedTensor as NestedTensorMulti  # noqa: E402

from tests._cases_detector import run_detector_cases  # noqa: E402

ns = SimpleNamespace(bsc=bsc, single=single, multipp=multipp, multi=multi, ts=ts, tpp=tpp, tm=tm, NestedTensor=NestedTensor, NestedTensorMulti=NestedTensorMulti,
                     PositionEmbeddingSine=PositionEmbeddingSine)
torch.set_grad_enabled(False)
blobs = {k: v.numpy() for k, v in run_detector_cases(ns).items()}
OUT = os.path.join(ROOT, "tests", "golden", "detector.npz")
np.savez_compressed(OUT, **blobs)
print("wrote", OUT, f"{os.path.getsize(OUT)/1e6:.2f} MB", len(blobs), "arrays")
for k, v in blobs.items():
    print(f"  {k:32s} {str(v.dtype):8s} {v.shape}")
