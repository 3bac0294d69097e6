"""Can the per-micro-batch pipeline be captured in a HIP graph (torch.cuda.CUDAGraph)?  Probe + timing."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "depth-fusion-in-transformer-based-video-object-detection_amd"))
sys.path.insert(0, ROOT)
from bench import build  # noqa: E402
from models.clip_inference import ClipRunner  # noqa: E402

torch.backends.cuda.matmul.allow_tf32 = False
torch.backends.cudnn.allow_tf32 = False
F_ = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda")
model = build(dev, F_ - 1)
runner = ClipRunner(model, micro_batch=F_)
x = torch.randn(F_, 4, 800, 1333, device=dev)


def step():
    return runner(x)


for _ in range(3):
    ref = step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    step()
torch.cuda.synchronize()
print(f"eager  : {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms per {F_}-frame step", flush=True)

g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
try:
    with torch.cuda.graph(g):
        out = step()
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    err = (out["pred_logits"] - ref["pred_logits"]).abs().max().item()
    t0 = time.perf_counter()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    print(f"graph  : {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms per {F_}-frame step, max |diff| vs eager {err:.2e}", flush=True)
except Exception as e:  # noqa: BLE001
    print("capture failed:", type(e).__name__, str(e)[:400], flush=True)
