"""Generate operator-level golden vectors from the REFERENCE's own PyTorch path.

Run in the build container only (needs /root/reference):
    python tools/gen_golden_op.py
Writes tests/golden/msda_op.npz.  Expected outputs come from the reference's
ms_deform_attn_core_pytorch (models/ops/functions/ms_deform_attn_func.py:41-61);
gradients come from autograd through that same function in float64, which is
what models/ops/test.py:63-78 checks the CUDA backward against (gradcheck).

Cases
  testpy_f64 / testpy_f32 : the reference's own fixture, models/ops/test.py:21-60
                            (N,M,D=1,2,2; Lq,L,P=2,2,2; shapes (6,4),(3,2); seed 3)
  enc_l1, dec_l1          : the two production shape classes (L=1, M=8, D=32, P=4)
                            on a down-scaled 8x12 map
  ms_l4                   : 4 levels
  border                  : locations in [-0.4, 1.4] (zero padding / skip rule)
  odd                     : M=3, D=5, P=3, L=2 (generic path)
  flatquirk               : loc shaped [1,Lq,8,R,4,2] read flat with L=1
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ref_import  # noqa: E402

fmod = ref_import.install()
core = fmod.ms_deform_attn_core_pytorch
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "msda_op.npz")


def lsi_of(shapes):
    return torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))


def make_case(seed, N, M, D, Lq, P, shape_list, dtype, lo=0.0, hi=1.0, with_grad=True):
    g = torch.Generator().manual_seed(seed)
    shapes = torch.as_tensor(shape_list, dtype=torch.long)
    L = shapes.shape[0]
    S = int(shapes.prod(1).sum())
    value = torch.randn(N, S, M, D, generator=g, dtype=torch.float64)
    loc = torch.rand(N, Lq, M, L, P, 2, generator=g, dtype=torch.float64) * (hi - lo) + lo
    aw = torch.rand(N, Lq, M, L, P, generator=g, dtype=torch.float64) + 1e-5
    aw = aw / aw.sum(-1, keepdim=True).sum(-2, keepdim=True)
    value, loc, aw = value.to(dtype), loc.to(dtype), aw.to(dtype)
    case = dict(value=value, shapes=shapes, lsi=lsi_of(shapes), loc=loc, aw=aw)
    with torch.no_grad():
        case["out"] = core(value, shapes, loc, aw)
    if with_grad:
        v, l, a = (t.detach().clone().double().requires_grad_(True) for t in (value, loc, aw))
        go = torch.randn(N, Lq, M * D, generator=g, dtype=torch.float64)
        core(v, shapes, l, a).backward(go)
        case.update(grad_out=go, grad_value=v.grad, grad_loc=l.grad, grad_aw=a.grad)
    return case


def main():
    blobs = {}

    def put(name, case):
        for k, v in case.items():
            blobs[f"{name}.{k}"] = v.detach().numpy()

    # --- the reference's own fixture, drawn exactly as models/ops/test.py does ---
    N, M, D, Lq, L, P = 1, 2, 2, 2, 2, 2
    shapes = torch.as_tensor([(6, 4), (3, 2)], dtype=torch.long)
    S = int(shapes.prod(1).sum())
    torch.manual_seed(3)
    for tag, dt in (("testpy_f64", torch.float64), ("testpy_f32", torch.float32)):
        value = torch.rand(N, S, M, D) * 0.01
        loc = torch.rand(N, Lq, M, L, P, 2)
        aw = torch.rand(N, Lq, M, L, P) + 1e-5
        aw /= aw.sum(-1, keepdim=True).sum(-2, keepdim=True)
        with torch.no_grad():
            out = core(value.to(dt), shapes, loc.to(dt), aw.to(dt))
        put(tag, dict(value=value.to(dt), shapes=shapes, lsi=lsi_of(shapes), loc=loc.to(dt), aw=aw.to(dt), out=out))

    put("enc_l1", make_case(11, 2, 8, 32, 96, 4, [(8, 12)], torch.float32, with_grad=False))
    put("dec_l1", make_case(12, 2, 8, 32, 30, 4, [(8, 12)], torch.float32, with_grad=False))
    put("ms_l4", make_case(13, 1, 8, 32, 24, 4, [(6, 8), (3, 4), (2, 2), (1, 1)], torch.float32))
    put("border", make_case(14, 1, 8, 32, 32, 4, [(5, 6)], torch.float32, lo=-0.4, hi=1.4))
    put("odd", make_case(15, 2, 3, 5, 17, 3, [(5, 4), (3, 2)], torch.float64))
    put("enc_l1_f64", make_case(16, 1, 8, 32, 20, 4, [(4, 5)], torch.float64))

    # --- flat-indexing quirk of the temporal decoder ---
    g = torch.Generator().manual_seed(17)
    R, Lq = 3, 30
    shapes = torch.as_tensor([(9, 11)], dtype=torch.long)
    value = torch.randn(1, 99, 8, 32, generator=g)
    loc = torch.rand(1, Lq, 8, R, 4, 2, generator=g)
    aw = torch.softmax(torch.randn(1, Lq, 8, 4, generator=g), -1).view(1, Lq, 8, 1, 4)
    with torch.no_grad():
        out = fmod.FlatShim.apply(value, shapes, lsi_of(shapes), loc, aw, 64)
    put("flatquirk", dict(value=value, shapes=shapes, lsi=lsi_of(shapes), loc=loc, aw=aw, out=out))

    np.savez_compressed(OUT, **blobs)
    print("wrote", os.path.normpath(OUT), f"{os.path.getsize(OUT)/1e3:.0f} kB", len(blobs), "arrays")


if __name__ == "__main__":
    main()
