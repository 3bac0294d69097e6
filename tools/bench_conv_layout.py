"""3x3 convolutions of ResNet-50 (800x1333, 8 frames) on MIOpen: NCHW vs channels_last tensors."""
import os
import sys
import time

import torch
import torch.nn.functional as F

torch.backends.cudnn.allow_tf32 = False
torch.backends.cudnn.benchmark = os.environ.get("BENCHMARK", "0") == "1"
dev = "cuda"
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8


def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


shapes = [("l1 3x3", 64, 64, 200, 334, 1, 1), ("l2 3x3 s2", 128, 128, 200, 334, 2, 1), ("l2 3x3", 128, 128, 100, 167, 1, 1),
          ("l3 3x3 s2", 256, 256, 100, 167, 2, 1), ("l3 3x3", 256, 256, 50, 84, 1, 1), ("l4 3x3 d2", 512, 512, 50, 84, 1, 2),
          ("stem 7x7", 3, 64, 800, 1333, 2, 1)]
for name, ci, co, h, w, st, dil in shapes:
    k = 7 if "7x7" in name else 3
    pad = 3 if k == 7 else dil
    x = torch.randn(N, ci, h, w, device=dev)
    wt = torch.randn(co, ci, k, k, device=dev) / (ci * k * k) ** 0.5
    ho, wo = (h + 2 * pad - dil * (k - 1) - 1) // st + 1, (w + 2 * pad - dil * (k - 1) - 1) // st + 1
    gf = 2 * ci * co * k * k * ho * wo * N / 1e9
    t1 = timeit(lambda: F.conv2d(x, wt, None, st, pad, dil))
    xc, wc = x.contiguous(memory_format=torch.channels_last), wt.contiguous(memory_format=torch.channels_last)
    t2 = timeit(lambda: F.conv2d(xc, wc, None, st, pad, dil))
    y = F.conv2d(xc, wc, None, st, pad, dil)
    print(f"{name:10s} {ci:4d}->{co:4d} {h}x{w}: NCHW {t1*1e6:8.1f} us {gf/t1/1e3:6.1f} TF | channels_last {t2*1e6:8.1f} us "
          f"{gf/t2/1e3:6.1f} TF  (out is channels_last: {y.is_contiguous(memory_format=torch.channels_last)})", flush=True)
