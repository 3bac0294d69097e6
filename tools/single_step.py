"""N forwards of the single-frame Late-Fusion detector at a given batch (for rocprofv3 --kernel-trace --stats):
    python tools/single_step.py [batch=1] [iters=20]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "depth-fusion-in-transformer-based-video-object-detection_amd"))
sys.path.insert(0, ROOT)
from models import build_model  # noqa: E402
from models.config import single_args  # noqa: E402
from models.fused import enable_fused_inference  # noqa: E402
from util.misc import nested_tensor_from_tensor_list  # noqa: E402

bs = int(sys.argv[1]) if len(sys.argv) > 1 else 1
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
torch.manual_seed(42)
model, _, _ = build_model(single_args("LateFusion", device="cuda"))
model = model.cuda().eval()
enable_fused_inference(model)
image = torch.randn(4, 800, 1333, generator=torch.Generator().manual_seed(1)).cuda()
inputs = nested_tensor_from_tensor_list([image] * bs)
with torch.no_grad():
    for _ in range(iters):
        model(inputs)
torch.cuda.synchronize()
