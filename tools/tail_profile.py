"""Runs only the query/RoI fusion + temporal stage (the 300-query tail) a few times, for rocprofv3."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "depth-fusion-in-transformer-based-video-object-detection_amd"))
sys.path.insert(0, ROOT)
from bench import build  # noqa: E402
from models.clip_inference import ClipRunner  # noqa: E402
from models.transformer_layers import make_level_tensors  # noqa: E402

torch.backends.cuda.matmul.allow_tf32 = False
F_, T, Q, C, S = int(os.environ.get("FRAMES", "8")), 32, 300, 256, 4200
dev = torch.device("cuda")
model = build(dev, T - 1)
runner = ClipRunner(model, micro_batch=F_)
tr = model.transformer
g = torch.Generator().manual_seed(0)
hs = torch.randn(F_, Q, C, generator=g).to(dev)
refs = torch.rand(F_, Q, 4, generator=g).to(dev) * 0.5 + 0.25
mem = torch.randn(F_, S, C, generator=g).to(dev)
pos = torch.randn(F_, S, C, generator=g).to(dev)
shapes, lsi = make_level_tensors([(50, 84)], dev)
whwh = torch.as_tensor((1333, 800, 1333, 800), dtype=torch.long, device=dev).repeat(1, Q, 1)
ratios = torch.ones(F_, 1, 2, device=dev)
mode = sys.argv[1] if len(sys.argv) > 1 else "both"
with torch.no_grad():
    for it in range(6):
        if mode in ("frame", "both"):
            fs = tr.frame_stage(hs, refs, mem, pos, (50, 84), whwh, model.class_embed[-1], model.bbox_embed[-1])
        else:
            fs = dict(cur=hs, ref=hs, logits=torch.randn(F_, Q, 3, device=dev))
        if mode in ("temporal", "both"):
            local = dict(cur=fs["cur"], ref_last=refs, memory=mem, spatial_shapes=shapes, level_start_index=lsi, valid_ratios=ratios)
            runner.temporal_forward(local, fs["ref"].repeat(T // F_, 1, 1), fs["logits"].repeat(T // F_, 1, 1), 0)
torch.cuda.synchronize()
