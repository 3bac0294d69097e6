"""fp32 GEMM microbench on the path's shapes: hand-written MFMA kernel vs the PyTorch library call."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "depth-fusion-in-transformer-based-video-object-detection_amd"))
from dfx import ops  # noqa: E402

torch.backends.cuda.matmul.allow_tf32 = False
torch.backends.cudnn.allow_tf32 = False


def timeit(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


F = int(os.environ.get('FRAMES', '8'))  # frames per micro-batch
LIB = os.environ.get('LIB', '1') == '1'
print("linear  [M,K] x [N,K]^T")
for name, M, N, K in (("value_proj", F * 4200, 256, 256), ("qproj", F * 4200, 96, 256), ("ffn1", F * 4200, 1024, 256),
                      ("ffn2", F * 4200, 256, 1024), ("dec 300", F * 300, 256, 256),
                      ("dyn layer", F * 300, 32768, 256), ("out_layer", F * 300, 256, 12544)):
    x = torch.randn(M, K, device="cuda")
    w = torch.randn(N, K, device="cuda") / K ** 0.5
    b = torch.randn(N, device="cuda")
    t0 = timeit(lambda: torch.nn.functional.linear(x, w, b))
    t1 = timeit(lambda: ops.linear(x, w, b))
    fl = 2.0 * M * N * K
    print(f"  {name:10s} M={M:6d} N={N:6d} K={K:6d}  torch {t0*1e6:8.1f} us {fl/t0/1e12:6.1f} TF | dfx {t1*1e6:8.1f} us {fl/t1/1e12:6.1f} TF", flush=True)
print("conv1x1 NCHW  W[Co,Ci] x X[Ci,HW]")
for name, Ci, Co, H, W in (("l1 c1", 256, 64, 200, 334), ("l1 c3", 64, 256, 200, 334), ("l2 c1", 512, 128, 100, 167),
                           ("l2 c3", 128, 512, 100, 167), ("l3 c1", 1024, 256, 50, 84), ("l3 c3", 256, 1024, 50, 84),
                           ("l4 c1", 2048, 512, 50, 84), ("l4 c3", 512, 2048, 50, 84), ("l4 down", 1024, 2048, 50, 84),
                           ("in_proj", 2048, 256, 50, 84)):
    x = torch.randn(F, Ci, H, W, device="cuda")
    w = torch.randn(Co, Ci, 1, 1, device="cuda") / Ci ** 0.5
    b = torch.randn(Co, device="cuda")
    res = torch.randn(F, Co, H, W, device="cuda") if "c3" in name else None
    t0 = timeit(lambda: torch.nn.functional.conv2d(x, w)) if LIB else float("nan")
    t1 = timeit(lambda: ops.conv1x1(x, w, b, residual=res, relu=True))
    fl = 2.0 * F * H * W * Ci * Co
    gb = 4.0 * F * H * W * (Ci + Co * (2 if res is not None else 1)) / 1e9
    print(f"  {name:8s} Ci={Ci:5d} Co={Co:5d} {H}x{W}  miopen(no epilogue) {t0*1e6:8.1f} us {fl/t0/1e12:6.1f} TF | dfx(+bias{'+res' if res is not None else ''}+relu) {t1*1e6:8.1f} us {fl/t1/1e12:6.1f} TF {gb/t1/1e3:5.2f} TB/s", flush=True)
