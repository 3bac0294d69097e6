"""Does running the early ResNet stages in small frame groups keep their activations in the 256 MB Infinity Cache?
python tools/mall_block.py   (32 frames, 800x1333; stem + layer1 + layer2 (+ layer3) in groups of g frames, then one torch.cat)
Layer1's 256-channel maps are 68 MB per frame: at 32 frames every tensor of the stage (2.2 GB) streams through HBM, in groups of
1-2 frames a bottleneck's working set fits the cache and the caching allocator hands the same addresses out again."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "depth-fusion-in-transformer-based-video-object-detection_amd"))
sys.path.insert(0, ROOT)
from bench import build  # noqa: E402
from models.fused import enable_fused_inference  # noqa: E402

dev = torch.device("cuda")
model = build(dev, 31)
enable_fused_inference(model, True)
body = model.backbone[0].body
x = torch.randn(32, 4, 800, 1333, device=dev)
stages = {"l1": [body.layer1], "l2": [body.layer1, body.layer2], "l3": [body.layer1, body.layer2, body.layer3]}


@torch.no_grad()
def run(upto, g):
    outs = []
    for s in range(0, 32, g):
        h = body.stem(x[s:s + g, :3], True)
        for st in stages[upto]:
            h = body.run_stage(st, h, True)
        outs.append(h)
    return outs[0] if len(outs) == 1 else torch.cat(outs, 0)


def timeit(fn, n=3):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


for upto in ("l1", "l2", "l3"):
    ref = run(upto, 32)
    for g in (32, 16, 8, 4, 2, 1):
        out = run(upto, g)
        same = torch.equal(out, ref)
        del out
        print(f"stem..{upto}  groups of {g:2d}: {timeit(lambda: run(upto, g)):8.2f} ms  bit-equal {same}", flush=True)
    del ref
