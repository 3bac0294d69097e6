"""What would the MSDA level kernel gain from (1) tap arithmetic shared between the 4 octet-workgroups of a head, (2) no LDS
bank conflicts?  Timing ablations (round 4; the outputs of builds 1 and 2 are WRONG by construction):

    python tools/level_ablate.py [spread]

Compiles csrc/msda_level.hip three times into libraries of their own (-DDFX_LEVEL_ABLATE=0 / 1 / 2; the shipped libdfx.so
carries neither ablation), runs the encoder geometry (50 x 84, block-major operands, per-query offsets of `spread` px) for
N = 32 / 8 / 4 frames, warm (operands in the Infinity Cache) and cold (384 MB written between launches), HIP events per launch.
  build 1: a thread computes the taps of its first query only (loads of the later queries' parameters kept): meant as the upper
           bound of what handing the taps over from octet 0 to the other three octet-workgroups could save - INCONCLUSIVE: hipcc
           spills 28-40 registers in this build (the shipped kernel sits at 125 of 128) and it runs slower than the shipped
           one; the instruction counts of the shipped ISA stand in: ~150 of the ~233 vector instructions per (query, octet)
           are tap arithmetic, 64 v_pk_fma_f32 + 32 ds_read_b128 + ~20 are the gather
  build 2: tap addresses replaced by lane-consecutive tokens: no LDS bank conflict
  build 3: no gather (nothing read from LDS), build 4: no staging loads - how the phases of an item add up"""
import ctypes
import os
import subprocess
import sys
import tempfile

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "depth-fusion-in-transformer-based-video-object-detection_amd")
sys.path.insert(0, PKG)
from dfx import _lib  # noqa: E402

spread = float(sys.argv[1]) if len(sys.argv) > 1 else 2.5
dev = torch.device("cuda:0")
H, W = 50, 84
S = H * W
P, I = ctypes.c_void_p, ctypes.c_int
libs = {}
for v in (0, 1, 2, 3, 4):
    so = os.path.join(tempfile.gettempdir(), f"liblevel_ablate{v}.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", f"-DDFX_LEVEL_ABLATE={v}",
                           "-I" + os.path.join(ROOT, "include"), "-shared", os.path.join(PKG, "csrc", "msda_level.hip"), "-o", so])
    lib = ctypes.CDLL(so)
    lib.dfx_msda_fused_level_forward_f32.argtypes = [P, P, I, P, P, P, I, I, I, I, P, P]
    libs[v] = lib
NAMES = {0: "shipped kernel", 1: "taps of a thread's first query only (WRONG results; spills: inconclusive)", 2: "lane-consecutive tap addresses: no bank conflicts (WRONG results)",
         3: "no gather: staging + tap arithmetic + stores only (WRONG results)", 4: "no staging loads: tap arithmetic + gather + stores only (WRONG results)"}
torch.manual_seed(0)
big = torch.empty(96 * 1024 * 1024, device=dev)
for N in (32, 8, 4):
    Lq = S
    value_blk = torch.randn(64, N * S, 4, device=dev)
    qproj = torch.randn(8, N * S, 12, device=dev)
    qproj[..., :8] *= spread
    ys, xs = torch.meshgrid((torch.arange(H) + 0.5) / H, (torch.arange(W) + 0.5) / W, indexing="ij")
    ref = torch.stack([xs.reshape(-1), ys.reshape(-1)], -1).view(1, S, 1, 2).expand(N, S, 1, 2).contiguous().to(dev)
    out = torch.empty(64, N * Lq, 4, device=dev)
    # the block-major layouts of dfx.ops.msda_level_forward (include/dfx_msda.h)
    rows = N * S
    ly = _lib.LevelLayout(S * 4, 4, 4 * rows * 8, 4 * rows * 2, 4 * rows, 12, rows * 12, 12, rows * 12, 4, 4 * rows * 8, 4 * rows * 2, 4 * rows)
    nbytes = 4 * (N * S * 256 + 3 * N * Lq * 32 + N * Lq * 256)
    for v, lib in libs.items():
        def run():
            rc = lib.dfx_msda_fused_level_forward_f32(value_blk.data_ptr(), ref.data_ptr(), 2, qproj.data_ptr(), qproj.data_ptr() + 32,
                                                      ctypes.byref(ly), N, H, W, Lq, out.data_ptr(), None)
            assert rc == 0
        for _ in range(3):
            run()
        res = {}
        for mode in ("warm", "cold"):
            ts = []
            for _ in range(15):
                if mode == "cold":
                    big.fill_(1.0)
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                run()
                b.record()
                torch.cuda.synchronize()
                ts.append(a.elapsed_time(b) * 1e-3)
            ts.sort()
            res[mode] = ts[len(ts) // 2]
        print(f"N={N:2d} build {v}: warm {res['warm'] * 1e6:7.1f} us ({nbytes / res['warm'] / 8e12:.3f} of 8 TB/s)   cold {res['cold'] * 1e6:7.1f} us "
              f"({nbytes / res['cold'] / 8e12:.3f})   {NAMES[v]}", flush=True)
