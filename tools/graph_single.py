"""Single-frame detector at batch 1 (the reference's benchmark.py default): eager launches vs one HIP-graph replay per image.
    python tools/graph_single.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "depth-fusion-in-transformer-based-video-object-detection_amd"))
sys.path.insert(0, ROOT)
from models import build_model  # noqa: E402
from models.config import single_args  # noqa: E402
from models.fused import enable_fused_inference  # noqa: E402
from util.misc import nested_tensor_from_tensor_list  # noqa: E402


def timed(fn, n=40):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return sum(ts[5:]) / len(ts[5:])


for fusion, channels in (("Baseline", 3), ("LateFusion", 4)):
    torch.manual_seed(42)
    model, _, _ = build_model(single_args(fusion, device="cuda"))
    model = model.cuda().eval()
    enable_fused_inference(model)
    for bs in (1, 2, 4):
        image = torch.randn(channels, 800, 1333, generator=torch.Generator().manual_seed(1)).cuda()
        inputs = nested_tensor_from_tensor_list([image] * bs)
        with torch.no_grad():
            eager = model(inputs)
            t_eager = timed(lambda: model(inputs))
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                model(inputs)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                out = model(inputs)
            graph.replay()
            torch.cuda.synchronize()
            same = torch.equal(out["pred_logits"], eager["pred_logits"]) and torch.equal(out["pred_boxes"], eager["pred_boxes"])
            t_graph = timed(graph.replay)
        print(f"{fusion:12s} batch {bs}: eager {t_eager * 1e3:6.2f} ms ({bs / t_eager:6.1f} FPS)   graph replay {t_graph * 1e3:6.2f} ms "
              f"({bs / t_graph:6.1f} FPS)   bit-equal {same}", flush=True)
