"""The 300-query Linears (M = 300 x frames) with forced tiles / the row kernel: python tools/bench_gemm_queries.py  (FRAMES, DFX_GEMM_TILE, DFX_GEMM_ROWS_MAX)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "depth-fusion-in-transformer-based-video-object-detection_amd"))
from dfx import ops  # noqa: E402


def timeit(fn, iters=40):
    for _ in range(5):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


F = int(os.environ.get("FRAMES", "32"))
M = 300 * F
for name, N, K, relu in (("dec ffn1", 1024, 256, True), ("dec ffn2", 256, 1024, False), ("proj 256", 256, 256, False),
                         ("kv 512", 512, 256, False), ("offs 768", 768, 256, False), ("ffn 2048", 2048, 256, True), ("ffn2 2048", 256, 2048, False)):
    x = torch.randn(M, K, device="cuda")
    w = torch.randn(N, K, device="cuda") / K ** 0.5
    b = torch.randn(N, device="cuda")
    t0 = timeit(lambda: torch.nn.functional.linear(x, w, b))
    t1 = timeit(lambda: ops.linear(x, w, b, relu=relu))
    fl = 2.0 * M * N * K
    print(f"  {name:10s} M={M:6d} N={N:6d} K={K:6d}  torch {t0*1e6:8.1f} us {fl/t0/1e12:6.1f} TF | dfx {t1*1e6:8.1f} us {fl/t1/1e12:6.1f} TF", flush=True)
