"""Generate module / transformer level golden vectors from the REFERENCE's own Python modules.

Build container only (imports /root/reference through tools/ref_import.py):
    python tools/gen_golden_models.py
Runs the shared recipes of tests/_cases.py against the reference's modules and writes their
outputs to tests/golden/models.npz.  Neither weights
nor inputs are stored: both are regenerated from seeds / state_dict names on the test side.

The MSDA operator inside the reference modules is the reference's ms_deform_attn_core_pytorch
behind the CUDA launcher's flat re-indexing (tools/ref_import.py); RoIAlign (third-party mmcv,
unpinned) is this repository's oracle restatement on both sides.
"""
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


def roi_fn(x, rois, output_size, spatial_scale, sampling_ratio, aligned):
    return msda_oracle.roi_align(x, rois, output_size, spatial_scale, sampling_ratio, aligned)


ref_import.install(roi_align_fn=roi_fn)
import models.deformable_transformer_multi as tm  # noqa: E402
import models.deformable_transformer_multi_plusplus as tpp  # noqa: E402
import models.deformable_transformer_single as ts  # noqa: E402
import models.dformer_backbone as dfb  # noqa: E402
import models.dformer_crossfusion_backbone as dcf  # noqa: E402
from models.ops.modules import MSDeformAttn  # noqa: E402
from models.position_encoding import PositionEmbeddingSine  # noqa: E402
from models.sparse_roi_head.head import RCNNHead  # noqa: E402
from util.misc import NestedTensor, inverse_sigmoid  # noqa: E402

from tests._cases import run_cases, run_train_cases  # noqa: E402

ns = SimpleNamespace(MSDeformAttn=MSDeformAttn, ts=ts, tpp=tpp, tm=tm, RCNNHead=RCNNHead, dfb=dfb, dcf=dcf,
                     PositionEmbeddingSine=PositionEmbeddingSine, NestedTensor=NestedTensor,
                     inverse_sigmoid=inverse_sigmoid)
torch.set_grad_enabled(False)
blobs = {k: v.numpy() for k, v in run_cases(ns).items()}
OUT = os.path.join(ROOT, "tests", "golden", "models.npz")
np.savez_compressed(OUT, **blobs)
print("wrote", OUT, f"{os.path.getsize(OUT)/1e6:.2f} MB", len(blobs), "arrays")

# the blocks in TRAIN mode under a fixed seed (sub-layer Dropouts active): tests/golden/train_mode.npz
blobs = {k: v.numpy() for k, v in run_train_cases(ns).items()}
OUT = os.path.join(ROOT, "tests", "golden", "train_mode.npz")
np.savez_compressed(OUT, **blobs)
print("wrote", OUT, f"{os.path.getsize(OUT)/1e6:.2f} MB", len(blobs), "arrays")
