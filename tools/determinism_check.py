"""Which building blocks give different bits run to run on this GPU?  (diagnostic)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "depth-fusion-in-transformer-based-video-object-detection_amd"))
import torch
import torch.nn.functional as F

from dfx import ops

torch.manual_seed(0)
dev = "cuda"


def check(name, fn, reps=5):
    outs = [fn().clone() for _ in range(reps)]
    torch.cuda.synchronize()
    diff = max((outs[0] - o).abs().max().item() for o in outs[1:])
    print(f"{name:40s} max run-to-run diff {diff:.3e}", flush=True)


for M, K, N in [(600, 256, 256), (2400, 256, 256), (2400, 256, 1024), (2400, 1024, 256), (8400, 256, 96), (300, 256, 91),
                (1200, 256, 768), (384, 256, 256), (12, 256, 256), (600, 512, 256), (33600, 256, 256)]:
    x, w, b = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev), torch.randn(N, device=dev)
    check(f"F.linear {M}x{K}x{N}", lambda: F.linear(x, w, b))
    check(f"dfx.linear {M}x{K}x{N}", lambda: ops.linear(x, w, b))
x = torch.randn(2, 300, 256, device=dev)
mha = torch.nn.MultiheadAttention(256, 8, batch_first=False).to(dev).eval()
with torch.no_grad():
    check("MultiheadAttention 300x2x256", lambda: mha(x.transpose(0, 1), x.transpose(0, 1), x.transpose(0, 1))[0])
    for shape, cin, cout, k, st, pad, dil in [((2, 64, 16, 24), 64, 64, 3, 1, 1, 1), ((2, 512, 4, 6), 512, 512, 3, 1, 2, 2),
                                              ((2, 4, 64, 96), 4, 64, 7, 2, 3, 1), ((2, 128, 8, 12), 128, 128, 3, 2, 1, 1),
                                              ((8, 256, 50, 84), 256, 256, 3, 1, 1, 1)]:
        xi = torch.randn(*shape, device=dev)
        w = torch.randn(cout, cin, k, k, device=dev)
        check(f"conv2d {shape} k{k} s{st} d{dil}", lambda: F.conv2d(xi, w, None, st, pad, dil))
    y = torch.randn(600, 256, device=dev)
    check("layer_norm", lambda: F.layer_norm(y, (256,)))
    check("softmax", lambda: torch.softmax(y, -1))
    a, bb = torch.randn(16, 300, 32, device=dev), torch.randn(16, 32, 300, device=dev)
    check("bmm 16x300x32x300", lambda: torch.bmm(a, bb))
