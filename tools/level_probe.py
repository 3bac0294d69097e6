"""Run the fused single-level MSDA launch of the encoder geometry (50 x 84, N frames) on the
wave-per-query kernel and on the level-in-LDS kernel, for rocprofv3 --kernel-trace --stats.

    rocprofv3 --kernel-trace --stats -d gpurun_out/level -- python tools/level_probe.py 8
"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..",
                                "depth-fusion-in-transformer-based-video-object-detection_amd"))
import torch

from dfx import ops
ops.LEVEL_ON_REFERENCE_LAYOUTS = True      # these probes compare the kernels on the reference layouts too
from models.transformer_layers import make_level_tensors


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    spread = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
    iters = 50
    load = len(sys.argv) > 3 and sys.argv[3] == "load"     # keep the matrix pipes busy between launches (DVFS)
    dev = torch.device("cuda:0")
    H, W, M, P = 50, 84, 8, 4
    S = H * W
    torch.manual_seed(0)
    shapes, lsi = make_level_tensors([(H, W)], dev)
    value = torch.randn(N, S, M, 32, device=dev)
    qproj = torch.randn(N, S, 3 * M * P, device=dev)
    qproj[..., : 2 * M * P] *= spread
    if spread == 0:     # the reference's initialisation: zero weight, bias = the same grid of offsets for every query
        qproj[..., : 2 * M * P] = (torch.randn(2 * M * P, device=dev) * 2.0)
    ys, xs = torch.meshgrid((torch.arange(H) + 0.5) / H, (torch.arange(W) + 0.5) / W, indexing="ij")
    ref = torch.stack([xs.reshape(-1), ys.reshape(-1)], -1).view(1, S, 1, 2).expand(N, S, 1, 2).contiguous().to(dev)
    value_blk = value.view(N, S, 64, 4).permute(2, 0, 1, 3).reshape(64, N * S, 4).contiguous()
    qproj_blk = torch.cat([qproj[..., :64].view(N, S, 8, 8), qproj[..., 64:].view(N, S, 8, 4)], -1) \
        .permute(2, 0, 1, 3).reshape(8, N * S, 12).contiguous()
    outs = {}
    for mode in ("wave", "level", "level_blk"):
        ops.USE_LEVEL_KERNEL = mode == "level"
        if mode == "level_blk":
            run = lambda: ops.msda_level_forward(value_blk, ref, qproj_blk, N, H, W)
        else:
            run = lambda: ops.msda_fused_forward(value, shapes, lsi, ref, qproj, 1, P)
        if load:
            a = torch.randn(4096, 4096, device=dev)
            for _ in range(iters):
                for _ in range(6):
                    a @ a
                out = run()
        for _ in range(iters):
            out = run()
        torch.cuda.synchronize()
        start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record()
        for _ in range(iters):
            out = run()
        stop.record()
        torch.cuda.synchronize()
        outs[mode] = out if mode != "level_blk" else out.view(64, N, S, 4).permute(1, 2, 0, 3).reshape(N, S, 256)
        print(f"{mode}: N={N} {start.elapsed_time(stop) / iters * 1e3:.1f} us per launch (back to back)", flush=True)
    print("max |level - wave|", (outs["level"] - outs["wave"]).abs().max().item(),
          " max |level_blk - level|", (outs["level_blk"] - outs["level"]).abs().max().item(), flush=True)

if __name__ == "__main__":
    main()
