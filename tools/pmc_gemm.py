"""A few GEMM launches for PMC passes (rocprofv3 --pmc ...): the hand-written kernel and the library on
the same shapes (conv1x1 of ResNet layer4 / layer1, the encoder projections and FFN)."""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "depth-fusion-in-transformer-based-video-object-detection_amd"))
from dfx import ops  # noqa: E402

torch.backends.cuda.matmul.allow_tf32 = False
torch.backends.cudnn.allow_tf32 = False
FR = int(os.environ.get("FRAMES", "8"))
for Ci, Co, H, W in ((1024, 2048, 50, 84), (512, 2048, 50, 84), (64, 256, 200, 334), (2048, 256, 50, 84)):
    x = torch.randn(FR, Ci, H, W, device="cuda")
    w = torch.randn(Co, Ci, 1, 1, device="cuda") / Ci ** 0.5
    b = torch.randn(Co, device="cuda")
    for _ in range(3):
        ops.conv1x1(x, w, b, relu=True)
    for _ in range(3):
        F.conv2d(x, w)
for M, N, K in ((4200 * FR, 256, 256), (4200 * FR, 1024, 256), (4200 * FR, 256, 1024), (4200 * FR, 96, 256), (300 * FR, 256, 12544)):
    x = torch.randn(M, K, device="cuda")
    w = torch.randn(N, K, device="cuda") / 16
    b = torch.randn(N, device="cuda")
    for _ in range(3):
        ops.linear(x, w, b)
    for _ in range(3):
        F.linear(x, w, b)
torch.cuda.synchronize()
