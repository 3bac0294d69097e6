"""Is a rank step bound by the host's launch rate?  For F frames per rank: wall time per step, the host time spent
enqueueing one step (no synchronisation inside), and the number of kernels our library launched.

    python tools/launch_bound.py [T=32]
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "depth-fusion-in-transformer-based-video-object-detection_amd"))
sys.path.insert(0, ROOT)
from bench import build  # noqa: E402
from dfx import ops  # noqa: E402
from models.clip_inference import ClipRunner  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda")
model = build(dev, T - 1)
clip = torch.randn(T, 4, 800, 1333, generator=torch.Generator().manual_seed(42))


def rank_step(runner, x):
    local = runner.frames_forward(x)
    rep = T // x.shape[0]
    return runner.temporal_forward(local, local["ref"].repeat(rep, 1, 1), local["logits"].repeat(rep, 1, 1), first_frame=0)


for F_ in (4, 8, 32):
    x = clip[:F_].to(dev)
    runner = ClipRunner(model, micro_batch=min(F_, 8), overlap=False)
    for _ in range(2):
        rank_step(runner, x)
    torch.cuda.synchronize()
    n = 4
    host = 0.0
    t0 = time.perf_counter()
    for _ in range(n):
        h0 = time.perf_counter()
        rank_step(runner, x)
        host += time.perf_counter() - h0
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / n
    ops.profile_start()
    rank_step(runner, x)
    torch.cuda.synchronize()
    rec = ops.profile_stop()
    busy = sum(r[0] for r in rec)
    print(f"frames/rank {F_:2d}: wall {wall * 1e3:7.2f} ms/step, host enqueue {host / n * 1e3:7.2f} ms/step, "
          f"own kernels {len(rec)} launches, {busy * 1e3:7.2f} ms stamped", flush=True)
