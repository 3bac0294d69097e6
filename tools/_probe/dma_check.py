import os, sys, torch
sys.path.insert(0, "/root/repo/depth-fusion-in-transformer-based-video-object-detection_amd")
from dfx import ops
torch.manual_seed(0)
for (M, N, K), tile in (((256, 256, 64), "5"), ((256, 256, 64), "2"), ((256, 256, 64), "0"), ((256, 256, 64), "6"), ((4200, 256, 256), None)):
    x = torch.randn(M, K).cuda(); w = torch.randn(N, K).cuda()
    if tile: os.environ["DFX_GEMM_TILE"] = tile
    else: os.environ.pop("DFX_GEMM_TILE", None)
    got = ops.linear(x, w)
    os.environ["DFX_GEMM_NO_DMA"] = "1"
    ref = ops.linear(x, w)
    os.environ.pop("DFX_GEMM_NO_DMA")
    d = (got - ref).abs()
    bad = (d > 1e-3)
    print(M, N, K, "tile", tile, "max diff", d.max().item(), "bad frac", bad.float().mean().item(),
          "bad rows", bad.any(1).nonzero().flatten()[:10].tolist(), "bad cols", bad.any(0).nonzero().flatten()[:10].tolist())
# conv form
x = torch.randn(2, 64, 16, 32).cuda(); w = torch.randn(128, 64, 1, 1).cuda()
got = ops.conv1x1(x, w)
os.environ["DFX_GEMM_NO_DMA"] = "1"; ref = ops.conv1x1(x, w); os.environ.pop("DFX_GEMM_NO_DMA")
d = (got - ref).abs(); print("conv1x1 max diff", d.max().item(), (d > 1e-3).float().mean().item())
