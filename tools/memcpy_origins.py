"""Which Python lines cause device-to-device memcpy launches (`__amd_rocclr_copyBuffer`) in a rank step?
    python tools/memcpy_origins.py [frames=4]
torch.profiler with stacks; prints the innermost frame inside this repository for every Memcpy DtoD event of one step."""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

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
pool_ref = torch.randn(T, 300, 256, device=dev)
pool_lg = torch.randn(T, 300, 3, device=dev)


def step():
    local = runner.frames_forward(x)
    return runner.temporal_forward(local, pool_ref, pool_lg, 0)


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
sites = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::cat", "aten::_to_copy", "aten::repeat", "aten::index", "aten::expand_as"):
        kids = " ".join(k.name for k in ev.cpu_children)
        dd = any("Memcpy" in k.name or "hipMemcpy" in k.name for k in ev.cpu_children) or "hipMemcpy" in kids
        if ev.name == "aten::copy_" and ("hipMemcpyAsync" in kids or "hipMemcpyWithStream" in kids):
            site = "?"
            for fr in ev.stack:
                if "depth-fusion-in-transformer" in fr and "tools/" not in fr:
                    site = fr.split("_amd/")[-1]
                    break
            sites[(site, tuple(ev.input_shapes[0]) if ev.input_shapes else None)] += 1
print(f"device-to-device memcpy launches in one {F_}-frame rank step: {sum(sites.values())}")
for (site, shape), c in sites.most_common(40):
    print(f"{c:4d}  {site}   {shape}")
