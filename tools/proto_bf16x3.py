"""PROTOTYPE measurement: an fp32 GEMM emulated with 9 (or 6) bf16 MFMA products
of operands split into three bf16 pieces, on ONE shape - layer4 conv1 of the ResNet-50 at 32 frames of 800x1333
(2048 -> 512 channels over 134 400 pixels) - against the shipped exact-fp32 MFMA GEMM (dfx.ops.linear):

    python tools/proto_bf16x3.py

compiles tools/proto/gemm_bf16x3.hip into a library of its own (nothing of it is in libdfx.so, no product route uses it),
times split + product, and prints the error of each variant against float64 on sampled rows.  Reported under its own
peak: the bf16 matrix pipe's 2.5 PFLOP/s / 9 (or / 6) products, not the fp32 pipe's 157.3 TFLOP/s."""
import ctypes
import os
import subprocess
import sys
import tempfile

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "depth-fusion-in-transformer-based-video-object-detection_amd"))
from dfx import ops  # noqa: E402

so = os.path.join(tempfile.gettempdir(), "libproto_bf16x3.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-shared",
                       os.path.join(ROOT, "tools", "proto", "gemm_bf16x3.hip"), "-o", so])
lib = ctypes.CDLL(so)
P, I, L = ctypes.c_void_p, ctypes.c_int, ctypes.c_long
lib.proto_split3.argtypes = [P, P, L, P]
lib.proto_gemm_bf16x3.argtypes = [P, P, P, I, I, I, I, P]


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


M, N, K = 4200 * 32, 512, 2048                          # pixels x output channels x input channels
g = torch.Generator(device="cuda").manual_seed(3)
x = torch.randn(M, K, device="cuda", generator=g).relu()      # post-ReLU activations
w = torch.randn(N, K, device="cuda", generator=g) / K ** 0.5
xs = torch.empty(3, M, K, dtype=torch.int16, device="cuda")
ws = torch.empty(3, N, K, dtype=torch.int16, device="cuda")
c = torch.empty(M, N, device="cuda")
st = torch.cuda.current_stream().cuda_stream
assert lib.proto_split3(w.data_ptr(), ws.data_ptr(), w.numel(), st) == 0          # weights: once per checkpoint
flops = 2.0 * M * N * K
t_split = timeit(lambda: lib.proto_split3(x.data_ptr(), xs.data_ptr(), x.numel(), st))
rows = torch.randint(0, M, (512,), device="cuda", generator=g)
want = x[rows].double() @ w.double().t()
scale = want.abs().max().item()
print(f"shape: [{M} x {K}] x [{N} x {K}]^T  (layer4 conv1, 32 frames), {flops / 1e12:.3f} TFLOP; outputs up to {scale:.2f}")
print(f"split of the activation operand into 3 x bf16: {t_split * 1e6:8.1f} us  ({x.numel() * 10 / t_split / 1e12:.2f} TB/s: 4 B read + 6 B written per element)")
t32 = timeit(lambda: ops.linear(x, w))
e32 = (ops.linear(x, w)[rows].double() - want).abs()
print(f"exact fp32 MFMA (dfx.ops.linear, shipped)      : {t32 * 1e6:8.1f} us  {flops / t32 / 1e12:6.1f} TFLOP/s = {flops / t32 / 157.3e12:.3f} of the fp32 matrix peak   "
      f"max err {e32.max().item():.2e}  rms {e32.pow(2).mean().sqrt().item():.2e}")
for nprod in (9, 6):
    fn = lambda: lib.proto_gemm_bf16x3(xs.data_ptr(), ws.data_ptr(), c.data_ptr(), M, N, K, nprod, st)  # noqa: E731
    assert fn() == 0
    t = timeit(fn)
    err = (c[rows].double() - want).abs()
    peak = 2.5e15 / nprod
    print(f"bf16 x 3 split, {nprod} products (prototype)           : {t * 1e6:8.1f} us  {flops / t / 1e12:6.1f} TFLOP/s fp32-equivalent = {flops / t / peak:.3f} of "
          f"2.5 PFLOP/s / {nprod}   max err {err.max().item():.2e}  rms {err.pow(2).mean().sqrt().item():.2e}   "
          f"with the split pass: {flops / (t + t_split) / 1e12:6.1f} TFLOP/s")
