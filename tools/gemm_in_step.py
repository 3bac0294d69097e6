"""The GEMM family of one 32-frame step, shape by shape: launches, time, TFLOP/s, share of the family's time.

    python tools/gemm_in_step.py [frames]

`dfx.ops.linear / conv1x1 / conv1x1_pair` are wrapped to note the shape of every call; the library's own launch stamps
(dfx_profile_*, the ones bench.py's `roofline` uses) come back in launch order and are matched one to one.  One clip at a
time on one stream (`bench.py --pipeline 0`'s schedule).
"""
import collections
import inspect
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "depth-fusion-in-transformer-based-video-object-detection_amd"))
sys.path.insert(0, ROOT)
from bench import build, FP32_MFMA_PEAK  # noqa: E402
from dfx import ops  # noqa: E402
from models.clip_inference import ClipRunner  # noqa: E402

F_ = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda")
model = build(dev, F_ - 1)
runner = ClipRunner(model, micro_batch=F_)
x = torch.randn(F_, 4, 800, 1333, device=dev)
calls = []
EACH = os.environ.get("EACH")          # "M,N,K": every launch of that conv1x1 shape on its own line, with the operands' addresses
where = []
_linear, _conv1x1, _pair = ops.linear, ops.conv1x1, ops.conv1x1_pair


def linear(*a, **k):
    b = inspect.signature(_linear).bind(*a, **k)
    b.apply_defaults()
    p = b.arguments
    x, weight = p["x"], p["weight"]
    if p["norm"] is None or ops._FUSE_LN:          # (an unfused norm= call comes back through here for its GEMM)
        M, K = (x.shape[1], x.shape[0] * 4) if p["x_blocked"] else (x.numel() // x.shape[-1], x.shape[-1])
        extra = "".join(t for t, on in (("+add", p["add"] is not None), ("+res", p["residual"] is not None),
                                        ("+relu", p["relu"] or p["act"] == "relu"), ("+gelu", p["act"] == "gelu"),
                                        ("+mask", p["row_mask"] is not None), (" blk-in", p["x_blocked"]),
                                        (" blk-out", bool(p["col_block"])), ("+ln", p["norm"] is not None)) if on)
        calls.append(("linear", M, weight.shape[0], K, extra))
    return _linear(*a, **k)


def conv1x1(x, weight, bias=None, residual=None, relu=False, stride=1):
    n, ci, h, w = x.shape
    h, w = (h + stride - 1) // stride, (w + stride - 1) // stride
    calls.append(("conv1x1", weight.shape[0], n * h * w, ci, ("+res" if residual is not None else "") + ("+relu" if relu else "")
                  + (f" /{stride}" if stride > 1 else "")))
    out = _conv1x1(x, weight, bias, residual, relu, stride)
    if EACH:
        where.append((len(calls) - 1, x.data_ptr(), 0 if residual is None else residual.data_ptr(), out.data_ptr(), x.stride(), tuple(out.stride())))
    return out


def conv1x1_pair(x1, x2, weight, bias=None, relu=False):
    n, c1, h, w = x1.shape
    calls.append(("conv1x1 pair", weight.shape[0], n * h * w, c1 + x2.shape[1], "+relu" if relu else ""))
    return _pair(x1, x2, weight, bias, relu)


with torch.no_grad():
    for _ in range(2):
        runner(x)
    torch.cuda.synchronize()
    ops.linear, ops.conv1x1, ops.conv1x1_pair = linear, conv1x1, conv1x1_pair
    ops.profile_start()
    runner(x)
    torch.cuda.synchronize()
    rec = [(sec, work, ta, tb) for (sec, work, ta, tb) in ops.profile_stop() if ta in (-1, -2)]
if len(rec) != len(calls):
    print(f"warning: {len(rec)} stamped launches for {len(calls)} wrapped calls (split-K / row ranges launch more than one)")
rows = collections.OrderedDict()
i = i_call = 0
for c in calls:
    if i >= len(rec):
        break
    sec, work, ta, tb = rec[i]
    i += 1
    flops = 2.0 * c[1] * c[2] * c[3]
    while abs(work - flops) > 0.01 * flops and i < len(rec) and rec[i][1] != flops and work < flops:   # row ranges: sum the pieces
        sec, work = sec + rec[i][0], work + rec[i][1]
        i += 1
    if EACH and c[0] == "conv1x1" and ",".join(str(v) for v in c[1:4]) == EACH:
        w = next((v for v in where if v[0] == i_call), None)
        print(f"  call {i_call:4d} {c[4]:12s} {sec * 1e6:8.1f} us  {work / sec / 1e12:6.1f} TFLOP/s" + ("" if w is None else
              f"  x {w[1]:#x} r {w[2]:#x} y {w[3]:#x}  x.stride {w[4]} y.stride {w[5]}"))
    i_call += 1
    r = rows.setdefault(c + (tb,), [0, 0.0, 0.0])
    r[0] += 1; r[1] += sec; r[2] += work
tot = sum(r[1] for r in rows.values())
tw = sum(r[2] for r in rows.values())
print(f"{F_} frames: {sum(r[0] for r in rows.values())} launches, {tot * 1e3:.2f} ms, {tw / tot / 1e12:.1f} TFLOP/s = {tw / tot / FP32_MFMA_PEAK:.3f}")
print("kind            M       N       K   epilogue        tile     n      ms   share  TFLOP/s   frac   ms lost vs 0.85")
for (kind, M, N, K, extra, tile), (n, sec, work) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
    tf = work / sec / 1e12
    lost = sec - work / (0.85 * FP32_MFMA_PEAK)
    print(f"{kind:12s} {M:7d} {N:7d} {K:7d}   {extra:14s} {tile // 1000:3d}x{tile % 1000:<3d} {n:4d} {sec * 1e3:8.3f} {sec / tot:6.3f} {tf:8.1f} {tf * 1e12 / FP32_MFMA_PEAK:6.3f}   {lost * 1e3:6.3f}")
