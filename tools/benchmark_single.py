"""The reference's own inference benchmark (benchmark.py:32-62) on the single-frame detectors of BASELINE.json configs 1-3:
one synthetic 800x1333 image (RGB or RGB-D) replicated --batch_size times, model(inputs) with a device synchronisation
on both sides of every iteration, the first --warm_iters iterations dropped, FPS = batch_size / mean iteration time.
Weights are each module's seeded initialisation (no checkpoints offline); fp32; fused inference routes on.

    python tools/benchmark_single.py [--num_iters 60] [--warm_iters 5] [--batch_sizes 1,8,32]
"""
import argparse
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

torch.backends.cuda.matmul.allow_tf32 = False
torch.backends.cudnn.allow_tf32 = False


@torch.no_grad()
def measure_average_inference_time(model, inputs, num_iters, warm_iters):
    ts = []
    for it in range(num_iters):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        model(inputs)
        torch.cuda.synchronize()
        if it >= warm_iters:
            ts.append(time.perf_counter() - t0)
    return sum(ts) / len(ts)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--num_iters", type=int, default=60)
    ap.add_argument("--warm_iters", type=int, default=5)
    ap.add_argument("--batch_sizes", default="1,8,32")
    ap.add_argument("--height", type=int, default=800)
    ap.add_argument("--width", type=int, default=1333)
    a = ap.parse_args()
    for fusion, channels in (("Baseline", 3), ("LateFusion", 4), ("Encoder_CrossFusion", 4)):
        torch.manual_seed(42)
        model, _, _ = build_model(single_args(fusion, device="cuda"))
        model = model.cuda().eval()
        enable_fused_inference(model)
        image = torch.randn(channels, a.height, a.width, generator=torch.Generator().manual_seed(1)).cuda()
        for bs in (int(b) for b in a.batch_sizes.split(",")):
            inputs = nested_tensor_from_tensor_list([image for _ in range(bs)])
            t = measure_average_inference_time(model, inputs, a.num_iters if bs < 16 else max(10, a.num_iters // 4), a.warm_iters)
            print(f"Deformable-DETR single frame, {fusion:20s} {a.height}x{a.width} batch {bs:2d}: {t * 1e3:8.2f} ms/iter  "
                  f"Inference Speed: {bs / t:7.1f} FPS", flush=True)
        del model
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
