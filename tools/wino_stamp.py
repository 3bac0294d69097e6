"""Where do the cycles of a Winograd K-loop iteration go?  (diagnostic; round 3)
    python tools/wino_stamp.py
Compiles csrc/conv_wino.hip with -DDFX_WINO_STAMP into a library of its own (the shipped libdfx.so carries no stamp), runs the
layer3 (256 -> 256, 50x84) and layer4 (512 -> 512, dilation 2) convolutions on 32 frames and prints, for wave 0 ("early": loads
the next chunk before its MFMAs, transforms / stores after) and wave 4 ("late": transforms / stores first) of eight workgroups,
the mean shader-clock cycles between the stamps of an iteration (s_memtime; MI355X_MICROARCH.md: the stamps cost ~11 %)."""
import ctypes
import os
import subprocess
import sys
import tempfile

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "depth-fusion-in-transformer-based-video-object-detection_amd")
sys.path.insert(0, PKG)
so = os.path.join(tempfile.gettempdir(), "libwino_stamp.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-DDFX_WINO_STAMP",
                       "-I" + os.path.join(ROOT, "include"), "-shared", os.path.join(PKG, "csrc", "conv_wino.hip"), "-o", so])
lib = ctypes.CDLL(so)
P, I = ctypes.c_void_p, ctypes.c_int
lib.dfx_conv3x3_wino_f32.argtypes = [P, P, P, P] + [I] * 7 + [P]
lib.dfx_wino_weights_f32.argtypes = [P, P, P, I, I, P]
lib.dfx_wino_set_stamp_buffer.argtypes = [P]
NAMES = ["late waves: transform + store of the next chunk", "operand fragments read from LDS (issued and landed)",
         "weight DMA + next patch loads issued", "32 MFMAs issued", "early waves: transform + store of the next chunk",
         "s_waitcnt vmcnt(0)", "barrier", "loop back"]
for Ci, Co, H, W, dil in ((256, 256, 50, 84, 1), (512, 512, 50, 84, 2), (64, 64, 200, 334, 1)):
    N = 32
    x = torch.randn(N, Ci, H, W, device="cuda")
    w = torch.randn(Co, Ci, 3, 3, device="cuda") / (Ci * 9) ** 0.5
    u = torch.empty(16 * Co * Ci, device="cuda")
    y = torch.empty(N, Co, H, W, device="cuda")
    b = torch.zeros(Co, device="cuda")
    nchunk = Ci // 8
    stamps = torch.zeros(8 * 2 * nchunk * 8, dtype=torch.int64, device="cuda")
    assert lib.dfx_wino_weights_f32(w.data_ptr(), None, u.data_ptr(), Co, Ci, None) == 0
    lib.dfx_wino_set_stamp_buffer(None)
    for _ in range(3):
        lib.dfx_conv3x3_wino_f32(x.data_ptr(), u.data_ptr(), b.data_ptr(), y.data_ptr(), N, Ci, H, W, Co, dil, 1, None)
    torch.cuda.synchronize()
    lib.dfx_wino_set_stamp_buffer(stamps.data_ptr())
    lib.dfx_conv3x3_wino_f32(x.data_ptr(), u.data_ptr(), b.data_ptr(), y.data_ptr(), N, Ci, H, W, Co, dil, 1, None)
    torch.cuda.synchronize()
    s = stamps.view(8, 2, nchunk, 8).cpu().double()
    print(f"\nconv {Ci}->{Co} {H}x{W} dilation {dil}, {nchunk} chunks; mean cycles per iteration over 8 workgroups, chunks 2..{nchunk - 2}")
    for wv, label in ((0, "wave 0 (early)"), (1, "wave 4 (late)")):
        t = s[:, wv, 2:nchunk - 1]
        nxt = s[:, wv, 3:nchunk, 0]
        seg = [(t[..., i + 1] - t[..., i]).mean().item() for i in range(7)] + [(nxt - t[..., 7]).mean().item()]
        total = (nxt - t[..., 0]).mean().item()
        print(f"  {label}: iteration {total:7.0f} cycles (MFMA block alone = 2048 per wave, 4096 per SIMD)")
        for nm, v in zip(NAMES, seg):
            print(f"      {v:7.0f}  {nm}")
