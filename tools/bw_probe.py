"""Practical HBM streaming rates on this chip: read-only (sum), write-only (fill), copy (read + write), for sizes
beyond the 256 MB Infinity Cache; context for the roofline fractions in DESIGN.md."""
import time

import torch

dev = "cuda"


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


for mb in (64, 256, 1024, 4096):
    n = mb * 1024 * 1024 // 4
    a = torch.randn(n, device=dev)
    b = torch.empty_like(a)
    tr = timeit(lambda: a.sum())
    tw = timeit(lambda: b.fill_(1.0))
    tc = timeit(lambda: b.copy_(a))
    print(f"{mb:5d} MB: read {a.numel() * 4 / tr / 1e12:5.2f} TB/s  write {a.numel() * 4 / tw / 1e12:5.2f} TB/s  "
          f"copy {2 * a.numel() * 4 / tc / 1e12:5.2f} TB/s (read + write bytes)", flush=True)
