"""A few launches of each hand-written MFMA kernel (and the library on the same GEMM shapes) for PMC passes
(rocprofv3 --pmc ...), 32 frames: GEMM in both operand forms, Winograd, implicit GEMM.  tools/pmc_report.py tabulates."""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "depth-fusion-in-transformer-based-video-object-detection_amd"))
from dfx import ops  # noqa: E402

torch.backends.cuda.matmul.allow_tf32 = False
torch.backends.cudnn.allow_tf32 = False
FR = int(os.environ.get("FRAMES", "32"))
LIB = os.environ.get("LIB", "1") == "1"
for Ci, Co, H, W in ((1024, 2048, 50, 84), (512, 2048, 50, 84), (1024, 256, 50, 84), (64, 256, 200, 334)):
    x = torch.randn(FR, Ci, H, W, device="cuda")
    w = torch.randn(Co, Ci, 1, 1, device="cuda") / Ci ** 0.5
    b = torch.randn(Co, device="cuda")
    for _ in range(3):
        ops.conv1x1(x, w, b, relu=True)
    if LIB:
        for _ in range(3):
            F.conv2d(x, w)
for M, N, K in ((4200 * FR, 256, 256), (4200 * FR, 1024, 256), (4200 * FR, 256, 1024), (300 * FR, 32768, 256)):
    x = torch.randn(M, K, device="cuda")
    w = torch.randn(N, K, device="cuda") / 16
    b = torch.randn(N, device="cuda")
    for _ in range(3):
        ops.linear(x, w, b)
    if LIB:
        for _ in range(3):
            F.linear(x, w, b)
for Ci, Co, H, W, k, s, p, d in ((64, 64, 200, 334, 3, 1, 1, 1), (256, 256, 50, 84, 3, 1, 1, 1), (512, 512, 50, 84, 3, 1, 2, 2),
                                 (128, 128, 200, 334, 3, 2, 1, 1), (3, 64, 800, 1333, 7, 2, 3, 1)):
    x = torch.randn(FR, Ci, H, W, device="cuda")
    w = torch.randn(Co, Ci, k, k, device="cuda") / (Ci * k * k) ** 0.5
    plan = ops.ConvPlan(w, torch.randn(Co, device="cuda"), s, p, d, "relu")
    for _ in range(3):
        plan(x)
torch.cuda.synchronize()
