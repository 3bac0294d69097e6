"""Dispatches for the PMC passes (run under `rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace`):

  1. calibration: dfx bias_act on a tensor of known size (reads 4 B + writes 4 B per element,
     16-byte coalesced streaming - the access width the gfx950 FETCH_SIZE correction is stated for)
  2. the MSDA fused kernel at the encoder geometry, N frames, after the value map was just written
     by a GEMM-like producer (torch.randn_like) so it is in the state the model leaves it in.
Prints the algorithmic bytes of both so the counter rows can be compared.
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "depth-fusion-in-transformer-based-video-object-detection_amd"))
from dfx import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=8)
ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
dev = "cuda"
torch.manual_seed(0)
# 1. calibration: 64 x 256 x 4096 fp32 = 256 MiB read + 256 MiB written
x = torch.randn(64, 256, 4096, device=dev)
b = torch.randn(256, device=dev)
for _ in range(a.reps):
    ops.bias_act_(x, b, None, True)
print("calibration bias_act: read", x.numel() * 4, "write", x.numel() * 4, "bytes per launch")
# 2. MSDA fused, encoder geometry
N, S, M, D, L, P = a.frames, 4200, 8, 32, 1, 4
shapes = torch.tensor([[50, 84]], dtype=torch.long, device=dev)
lsi = torch.zeros(1, dtype=torch.long, device=dev)
ys, xs = torch.meshgrid(torch.linspace(0.5, 49.5, 50) / 50, torch.linspace(0.5, 83.5, 84) / 84, indexing="ij")
ref = torch.stack([xs.reshape(-1), ys.reshape(-1)], -1).view(1, S, 1, 2).expand(N, S, 1, 2).contiguous().to(dev)
qproj = torch.randn(N, S, 3 * M * L * P, device=dev)
qproj[..., : 2 * M * L * P] *= 3.0          # offsets of a few pixels
for _ in range(a.reps):
    value = torch.randn(N, S, M, D, device=dev)
    ops.msda_fused_forward(value, shapes, lsi, ref, qproj, L, P)
nb = 4 * (N * S * M * D + 3 * N * S * M * L * P + N * S * M * D)
print("msda_fused enc: algorithmic bytes per launch", nb, "(value", 4 * N * S * M * D, "qproj", 12 * N * S * M * L * P,
      "out", 4 * N * S * M * D, ")")
torch.cuda.synchronize()
# 3. the level-in-LDS kernel on the same problem, operands in the block-major layouts the model path uses
value_blk_shape, qb = (64, N * S, 4), torch.cat([qproj[..., :64].view(N, S, 8, 8), qproj[..., 64:].view(N, S, 8, 4)], -1) \
    .permute(2, 0, 1, 3).reshape(8, N * S, 12).contiguous()
for _ in range(a.reps):
    value_blk = torch.randn(value_blk_shape, device=dev)
    ops.msda_level_forward(value_blk, ref, qb, N, 50, 84)
print("msda_fused_level: same algorithmic bytes per launch", nb)
torch.cuda.synchronize()
