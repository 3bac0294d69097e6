"""Where does the HOST time of a rank step go (cProfile, GPU work not waited for)?  python tools/host_profile.py [frames=4]"""
import cProfile
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "depth-fusion-in-transformer-based-video-object-detection_amd"))
sys.path.insert(0, ROOT)
from bench import build  # noqa: E402
from models.clip_inference import ClipRunner  # noqa: E402

F_ = int(sys.argv[1]) if len(sys.argv) > 1 else 4
T = 32
dev = torch.device("cuda")
model = build(dev, T - 1)
runner = ClipRunner(model, micro_batch=F_)
x = torch.randn(F_, 4, 800, 1333, device=dev)


def step():
    local = runner.frames_forward(x)
    rep = T // F_
    return runner.temporal_forward(local, local["ref"].repeat(rep, 1, 1), local["logits"].repeat(rep, 1, 1), 0)


with torch.no_grad():
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(f"host enqueue {1e3 * (t1 - t0) / 5:.2f} ms per {F_}-frame step")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(5):
        step()
    pr.disable()
    torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
