"""The few-input-channel convolutions (ResNet stem 3 -> 64 7x7/2, DFormer 1 -> 16 3x3/2) on the tile kernel and on the implicit
GEMM, a few launches each, for PMC passes (rocprofv3 --pmc ...; tools/pmc_report.py tabulates).  FRAMES (32)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "depth-fusion-in-transformer-based-video-object-detection_amd"))
from dfx import ops  # noqa: E402

FR = int(os.environ.get("FRAMES", "32"))
for Ci, Co, k, p in ((3, 64, 7, 3), (1, 16, 3, 1)):
    x = torch.randn(FR, Ci, 800, 1333, device="cuda")
    w = torch.randn(Co, Ci, k, k, device="cuda") / (Ci * k * k) ** 0.5
    b = torch.randn(Co, device="cuda")
    for algo in ("tile", "igemm"):
        plan = ops.ConvPlan(w, b, 2, p, 1, "relu", algo=algo)
        for _ in range(4):
            plan(x)
torch.cuda.synchronize()
