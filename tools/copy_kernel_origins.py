"""Where do the small elementwise / copy kernels of a rank step come from?  torch.profiler with Python
stacks, grouped by the innermost frame inside this repository.

    python tools/copy_kernel_origins.py [frames=4]
"""
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


def step():
    local = runner.frames_forward(x)
    rep = T // F_
    return runner.temporal_forward(local, local["ref"].repeat(rep, 1, 1), local["logits"].repeat(rep, 1, 1), 0)


for _ in range(2):
    step()
torch.cuda.synchronize()
import traceback  # noqa: E402

from torch.overrides import TorchFunctionMode  # noqa: E402

WATCH = {"contiguous", "clone", "to", "cat", "stack", "repeat", "add", "mul", "div", "sub", "clamp", "sigmoid", "log",
         "masked_fill", "gather", "index_select", "__getitem__", "__setitem__", "flatten", "permute", "transpose",
         "reshape", "float", "type", "copy_", "expand", "where", "topk", "sort", "__add__", "__mul__", "__truediv__",
         "__sub__", "__radd__", "__rmul__", "relu", "softmax", "layer_norm", "linear", "exp", "neg", "zeros", "ones",
         "full", "arange", "as_tensor", "tensor", "__rsub__", "unbind", "split", "chunk", "sum", "mean", "max", "min"}
count = collections.Counter()


class Spy(TorchFunctionMode):
    def __torch_function__(self, func, types, args=(), kwargs=None):
        name = getattr(func, "__name__", str(func))
        if name in WATCH:
            site = "?"
            for fr in reversed(traceback.extract_stack(limit=12)[:-1]):
                if "depth-fusion-in-transformer" in fr.filename and "tools/" not in fr.filename:
                    site = fr.filename.split("_amd/")[-1] + ":" + str(fr.lineno)
                    break
            count[(name, site)] += 1
        return func(*args, **(kwargs or {}))


with Spy():
    step()
    torch.cuda.synchronize()
tot = collections.Counter()
for (name, site), c in count.items():
    tot[site] += c
print("calls per site (all watched ops):")
for site, c in tot.most_common(45):
    names = ", ".join(f"{n}x{k}" for (n, s2), k in count.items() if s2 == site)
    print(f"{c:5d}  {site:60s} {names[:120]}")

# ---- device-to-device copies the runtime serves with its own copy kernel (`__amd_rocclr_copyBuffer` in a rocprofv3 trace):
# ATen copy_ / clone / contiguous of same-dtype, densely laid-out operands go through hipMemcpyAsync ----
from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402

copies = collections.Counter()


class CopySpy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        if any(k in name for k in ("copy_", "clone", "_to_copy", "contiguous")):
            src = args[1] if "copy_" in name and len(args) > 1 else args[0]
            dst = args[0] if "copy_" in name else out
            if (isinstance(src, torch.Tensor) and isinstance(dst, torch.Tensor) and src.is_cuda and dst.is_cuda
                    and src.dtype == dst.dtype and src.is_contiguous() and dst.is_contiguous() and src.numel() == dst.numel()):
                site = "?"
                for fr in reversed(traceback.extract_stack(limit=16)[:-1]):
                    if "depth-fusion-in-transformer" in fr.filename and "tools/" not in fr.filename:
                        site = fr.filename.split("_amd/")[-1] + ":" + str(fr.lineno)
                        break
                copies[(name, site, tuple(src.shape))] += 1
        return out


with CopySpy():
    step()
    torch.cuda.synchronize()
print(f"\ndense same-dtype device copies (memcpy path) in one {F_}-frame rank step: {sum(copies.values())}")
for (name, site, shape), c in copies.most_common(40):
    print(f"{c:5d}  {name:28s} {site:58s} {shape}")
