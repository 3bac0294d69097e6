"""Kernel-level timing of the MSDA forward at the production shape classes (GPU box).

python tools/bench_msda.py [--frames 32] [--iters 200]
Prints algorithmic GB/s per call (value + loc/aw + out bytes) for
  enc  : N=frames, Lq=S=4200, L=1      dec : N=frames, Lq=300, S=4200, L=1
  enc4 : N=frames/4, Lq=S=22223, L=4   unfused vs fused front end.
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "depth-fusion-in-transformer-based-video-object-detection_amd"))
import MultiScaleDeformableAttention as MSDA  # noqa: E402
from dfx import ops
ops.LEVEL_ON_REFERENCE_LAYOUTS = True      # these probes compare the kernels on the reference layouts too  # noqa: E402


def timeit(fn, iters):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=32)
    ap.add_argument("--iters", type=int, default=200)
    a = ap.parse_args()
    torch.manual_seed(42)
    dev = "cuda"
    M, D, P = 8, 32, 4
    for name, N, Lq, shp, realistic in (
            ("enc_uniform", a.frames, 4200, [(50, 84)], False), ("enc_grid", a.frames, 4200, [(50, 84)], True),
            ("enc_grid_n1", 1, 4200, [(50, 84)], True), ("enc_grid_n4", 4, 4200, [(50, 84)], True),
            ("dec", a.frames, 300, [(50, 84)], False),
            ("enc_L4", max(1, a.frames // 4), 22223, [(100, 167), (50, 84), (25, 42), (13, 21)], False)):
        shapes = torch.as_tensor(shp, dtype=torch.long, device=dev)
        lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
        L, S = len(shp), int(shapes.prod(1).sum())
        value = torch.randn(N, S, M, D, device=dev)
        if realistic and L == 1:
            H, W = shp[0]
            ys, xs = torch.meshgrid(torch.linspace(0.5, H - 0.5, H) / H, torch.linspace(0.5, W - 0.5, W) / W, indexing="ij")
            ref = torch.stack([xs.reshape(-1), ys.reshape(-1)], -1).to(dev)          # [S,2]
            loc = ref[None, :, None, None, None, :] + torch.randn(N, Lq, M, L, P, 2, device=dev) * (4.0 / W)
        else:
            loc = torch.rand(N, Lq, M, L, P, 2, device=dev)
        aw = torch.softmax(torch.randn(N, Lq, M, L * P, device=dev), -1).view(N, Lq, M, L, P)
        loc = loc.contiguous()
        nbytes = 4 * (N * S * M * D + 3 * N * Lq * M * L * P + N * Lq * M * D)
        t = timeit(lambda: MSDA.ms_deform_attn_forward(value, shapes, lsi, loc, aw, 64), a.iters)
        line = f"{name:12s} N={N:3d} Lq={Lq:6d} L={L}  unfused {t*1e6:8.1f} us  {nbytes/t/1e9:8.1f} GB/s alg"
        # fused: reference points + raw projections
        qproj = torch.randn(N, Lq, 3 * M * L * P, device=dev)
        refp = torch.rand(N, Lq, L, 2, device=dev)
        tf = timeit(lambda: ops.msda_fused_forward(value, shapes, lsi, refp, qproj, L, P), a.iters)
        line += f" | fused {tf*1e6:8.1f} us  {nbytes/tf/1e9:8.1f} GB/s alg"
        if L == 1 and Lq == S:
            from models.transformer_layers import make_level_tensors
            sh2, lsi2 = make_level_tensors(shp, dev)
            H, W = shp[0]
            ys, xs = torch.meshgrid((torch.arange(H) + 0.5) / H, (torch.arange(W) + 0.5) / W, indexing="ij")
            grid = torch.stack([xs.reshape(-1), ys.reshape(-1)], -1).view(1, S, 1, 2).expand(N, S, 1, 2).contiguous().to(dev)
            for spread in (1.0, 3.0):
                q2 = qproj.clone()
                q2[..., : 2 * M * L * P] *= spread
                for mode in ("wave", "level"):
                    ops.USE_LEVEL_KERNEL = mode == "level"
                    tt = timeit(lambda: ops.msda_fused_forward(value, sh2, lsi2, grid, q2, L, P), a.iters)
                    line += f" | sd{spread:.0f}px {mode} {tt*1e6:7.1f} us {nbytes/tt/1e9:7.1f} GB/s"
                ops.USE_LEVEL_KERNEL = True
        print(line, flush=True)


if __name__ == "__main__":
    main()
