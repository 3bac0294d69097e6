"""Clips as a stream with R independent ClipRunner pipelines (2 HIP streams each) fed round-robin: does more
concurrency than the default single pipeline (2 clips in flight) fill the chip better?
    python tools/multi_runner.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "depth-fusion-in-transformer-based-video-object-detection_amd"))
sys.path.insert(0, ROOT)
from bench import build  # noqa: E402
from models.clip_inference import ClipRunner  # noqa: E402

T = 32
dev = torch.device("cuda")
model = build(dev, T - 1)
clip = torch.randn(T, 4, 800, 1333, generator=torch.Generator().manual_seed(42)).to(dev)
for R in (1, 2, 3):
    runners = [ClipRunner(model, micro_batch=T, overlap=False) for _ in range(R)]
    for i in range(3 * R):
        runners[i % R].submit(clip)
    torch.cuda.synchronize()
    n = 12
    t0 = time.perf_counter()
    for i in range(n):
        runners[i % R].submit(clip)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"{R} pipeline(s): {dt * 1e3:7.2f} ms per clip -> {T / dt:6.1f} frames/s", flush=True)
    del runners
    torch.cuda.empty_cache()
