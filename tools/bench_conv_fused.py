"""3x3 convolution + bias + ReLU: library convolution followed by dfx bias_act_ vs torch.miopen_convolution_relu."""
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "depth-fusion-in-transformer-based-video-object-detection_amd"))
from dfx import ops  # noqa: E402

torch.backends.cudnn.allow_tf32 = False
dev = "cuda"


def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


print("has miopen_convolution_relu:", hasattr(torch, "miopen_convolution_relu"))
for name, ci, co, h, w, st, dil in [("l1", 64, 64, 200, 334, 1, 1), ("l2", 128, 128, 100, 167, 1, 1), ("l3", 256, 256, 50, 84, 1, 1),
                                    ("l4", 512, 512, 50, 84, 1, 2)]:
    x = torch.randn(8, ci, h, w, device=dev)
    wt = torch.randn(co, ci, 3, 3, device=dev) / (ci * 9) ** 0.5
    b = torch.randn(co, device=dev)
    t1 = timeit(lambda: ops.bias_act_(F.conv2d(x, wt, None, st, dil, dil), b, relu=True))
    t0 = timeit(lambda: F.conv2d(x, wt, None, st, dil, dil))
    try:
        t2 = timeit(lambda: torch.miopen_convolution_relu(x, wt, b, [st, st], [dil, dil], [dil, dil], 1))
        y2 = torch.miopen_convolution_relu(x, wt, b, [st, st], [dil, dil], [dil, dil], 1)
        y1 = ops.bias_act_(F.conv2d(x, wt, None, st, dil, dil), b, relu=True)
        err = (y1 - y2).abs().max().item()
    except Exception as e:  # noqa: BLE001
        t2, err = float("nan"), str(e)[:80]
    print(f"{name}: conv {t0*1e6:7.1f} us | conv + bias_act {t1*1e6:7.1f} us | miopen_convolution_relu {t2*1e6:7.1f} us  max diff {err}")
