"""Stamped duration of the level-in-LDS MSDA kernel (block-major operands, encoder geometry 50 x 84) for N frames with
per-query offsets of `spread` px:  python tools/level_time.py [spread]   (DFX_LEVEL_NOT_PERSISTENT=1: one workgroup per item)"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "depth-fusion-in-transformer-based-video-object-detection_amd"))
import torch

from dfx import ops

spread = float(sys.argv[1]) if len(sys.argv) > 1 else 2.5
dev = torch.device("cuda:0")
H, W = 50, 84
S = H * W
torch.manual_seed(0)
for N in (32, 8, 4):
    value_blk = torch.randn(64, N * S, 4, device=dev)
    qproj = torch.randn(8, N * S, 12, device=dev)
    qproj[..., :8] *= spread
    ys, xs = torch.meshgrid((torch.arange(H) + 0.5) / H, (torch.arange(W) + 0.5) / W, indexing="ij")
    ref = torch.stack([xs.reshape(-1), ys.reshape(-1)], -1).view(1, S, 1, 2).expand(N, S, 1, 2).contiguous().to(dev)
    big = torch.empty(96 * 1024 * 1024, device=dev)             # 384 MB: evicts the operands from the Infinity Cache
    for _ in range(3):
        ops.msda_level_forward(value_blk, ref, qproj, N, H, W)
    torch.cuda.synchronize()
    ops.profile_start()
    for _ in range(20):
        big.fill_(1.0)
        ops.msda_level_forward(value_blk, ref, qproj, N, H, W)
    torch.cuda.synchronize()
    rec = ops.profile_stop()
    t = sorted(r[0] for r in rec)
    nbytes = rec[0][1]
    med = t[len(t) // 2]
    print(f"N={N:2d}  median {med * 1e6:7.2f} us  min {t[0] * 1e6:7.2f} us  algorithmic {nbytes / 1e6:.1f} MB -> {nbytes / med / 1e12:.2f} TB/s = {nbytes / med / 8e12:.3f} of 8 TB/s", flush=True)
