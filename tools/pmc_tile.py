import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "depth-fusion-in-transformer-based-video-object-detection_amd"))
from dfx import ops
from models.transformer_layers import make_level_tensors
from models.ops.modules import MSDeformAttn
N, H, W = 8, 50, 84
S = H * W
dev = "cuda"
sh, lsi = make_level_tensors([(H, W)], dev)
value = torch.randn(N, S, 8, 32, device=dev)
ys, xs = torch.meshgrid((torch.arange(H) + 0.5) / H, (torch.arange(W) + 0.5) / W, indexing="ij")
grid = torch.stack([xs.reshape(-1), ys.reshape(-1)], -1).view(1, S, 1, 2).expand(N, S, 1, 2).contiguous().to(dev)
m = MSDeformAttn(256, 1, 8, 4)                      # the module's own initial offsets: k * direction, k = 1..4
q = torch.cat([m.sampling_offsets.bias.detach().view(1, 1, 64).expand(N, S, 64), torch.zeros(N, S, 32)], -1).contiguous().to(dev)
for flag in (False, True):
    ops.USE_TILE_KERNEL = flag
    for _ in range(3):
        ops.msda_fused_forward(value, sh, lsi, grid, q, 1, 4)
torch.cuda.synchronize()
