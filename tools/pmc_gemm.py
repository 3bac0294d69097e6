"""A few GEMM launches for PMC passes (rocprofv3 --pmc ...): conv1x1 shapes of ResNet layer4 / layer1."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "depth-fusion-in-transformer-based-video-object-detection_amd"))
from dfx import ops  # noqa: E402

for Ci, Co, H, W in ((1024, 2048, 50, 84), (512, 2048, 50, 84), (64, 256, 200, 334)):
    x = torch.randn(8, Ci, H, W, device="cuda")
    w = torch.randn(Co, Ci, 1, 1, device="cuda") / Ci ** 0.5
    b = torch.randn(Co, device="cuda")
    for _ in range(3):
        ops.conv1x1(x, w, b, relu=True)
x = torch.randn(33600, 256, device="cuda")
w = torch.randn(1024, 256, device="cuda") / 16
for _ in range(3):
    ops.linear(x, w, None, relu=True)
torch.cuda.synchronize()
