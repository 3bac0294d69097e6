"""GPU parity: the HIP operator (through the C ABI) against the golden vectors and the CPU oracle.

Tolerances: fp32 attention outputs within 1e-3 absolute (BASELINE.json north_star); we assert a
tighter 2e-5 * scale here.  fp64 within the reference's own allclose defaults (test.py:39).
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

F32_TOL = dict(rtol=1e-4, atol=2e-5)
F64_TOL = dict(rtol=1e-5, atol=1e-8)


@pytest.fixture(scope="module")
def msda():
    import MultiScaleDeformableAttention as MSDA
    from dfx import _lib
    _lib.load()  # fail loudly if the HIP library is missing
    return MSDA


@pytest.fixture(scope="module")
def z(golden_dir):
    return np.load(os.path.join(golden_dir, "msda_op.npz"))


def _t(z, case, key, dev="cuda"):
    return torch.from_numpy(z[f"{case}.{key}"]).to(dev)


def lsi_of(shapes):
    return torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))


def rand_case(seed, N, M, D, Lq, P, shape_list, dtype=torch.float32, lo=0.0, hi=1.0):
    g = torch.Generator().manual_seed(seed)
    shapes = torch.as_tensor(shape_list, dtype=torch.long)
    L, S = shapes.shape[0], int(shapes.prod(1).sum())
    value = torch.randn(N, S, M, D, generator=g, dtype=dtype)
    loc = (torch.rand(N, Lq, M, L, P, 2, generator=g, dtype=dtype) * (hi - lo) + lo)
    aw = torch.softmax(torch.randn(N, Lq, M, L * P, generator=g, dtype=dtype), -1).view(N, Lq, M, L, P)
    return value, shapes, lsi_of(shapes), loc, aw


def gpu(ts):
    return [t.cuda() for t in ts]


@pytest.mark.parametrize("case", ["testpy_f32", "enc_l1", "dec_l1", "ms_l4", "border", "flatquirk",
                                  "testpy_f64", "odd", "enc_l1_f64"])
def test_forward_matches_reference_vectors(msda, z, case):
    args = [_t(z, case, k) for k in ("value", "shapes", "lsi", "loc", "aw")]
    out = msda.ms_deform_attn_forward(*args, 64).cpu()
    ref = _t(z, case, "out", "cpu")
    assert out.shape == ref.shape and out.dtype == ref.dtype
    assert torch.allclose(out, ref, **(F64_TOL if out.dtype == torch.float64 else F32_TOL))


@pytest.mark.parametrize("case", ["ms_l4", "border", "odd", "enc_l1_f64"])
def test_backward_matches_reference_autograd(msda, z, case):
    args = [_t(z, case, k) for k in ("value", "shapes", "lsi", "loc", "aw", "grad_out")]
    args = [a.double() if a.is_floating_point() else a for a in args]
    gv, gl, ga = msda.ms_deform_attn_backward(*args, 64)
    assert torch.allclose(gv.cpu(), _t(z, case, "grad_value", "cpu"), rtol=1e-8, atol=1e-10)
    assert torch.allclose(gl.cpu(), _t(z, case, "grad_loc", "cpu"), rtol=1e-8, atol=1e-9)
    assert torch.allclose(ga.cpu(), _t(z, case, "grad_aw", "cpu"), rtol=1e-8, atol=1e-10)


GEOMS = [
    # (N, M, D, Lq, P, shapes, dtype, lo, hi)
    (2, 8, 32, 333, 4, [(20, 31)], torch.float32, 0.0, 1.0),          # fast path, L=1
    (1, 8, 32, 1001, 4, [(16, 20), (8, 10), (4, 5), (2, 3)], torch.float32, 0.0, 1.0),  # fast path, L=4
    (3, 8, 32, 50, 4, [(9, 7)], torch.float32, -0.5, 1.5),             # fast path, border rule
    (2, 8, 32, 77, 3, [(9, 7), (4, 3)], torch.float32, -0.1, 1.1),     # fast path, run-time P
    (2, 8, 32, 5, 4, [(1, 1)], torch.float32, -0.2, 1.2),              # 1x1 map
    (2, 4, 16, 40, 2, [(6, 5)], torch.float32, 0.0, 1.0),              # generic fp32
    (1, 3, 7, 19, 5, [(5, 4), (2, 2)], torch.float32, -0.2, 1.2),      # generic odd
    (2, 8, 32, 40, 4, [(6, 5)], torch.float64, -0.2, 1.2),             # fp64
    (1, 2, 71, 9, 2, [(6, 4), (3, 2)], torch.float64, 0.0, 1.0),       # test.py channel sweep member
]


@pytest.mark.parametrize("geom", GEOMS, ids=[str(i) for i in range(len(GEOMS))])
def test_forward_matches_oracle(msda, oracle, geom):
    N, M, D, Lq, P, shp, dt, lo, hi = geom
    args = rand_case(100 + Lq, N, M, D, Lq, P, shp, dt, lo, hi)
    ref = oracle.msda_forward(*args)
    out = msda.ms_deform_attn_forward(*gpu(args), 64).cpu()
    assert torch.allclose(out, ref, **(F64_TOL if dt == torch.float64 else F32_TOL))
    assert (out - ref).abs().max() < 1e-3        # the north-star bound, stated explicitly


@pytest.mark.parametrize("geom", GEOMS, ids=[str(i) for i in range(len(GEOMS))])
def test_backward_matches_oracle(msda, oracle, geom):
    N, M, D, Lq, P, shp, dt, lo, hi = geom
    args = rand_case(200 + Lq, N, M, D, Lq, P, shp, dt, lo, hi)
    go = torch.randn(N, Lq, M * D, generator=torch.Generator().manual_seed(5), dtype=dt)
    ref = oracle.msda_backward(*[a.double() if a.is_floating_point() else a for a in args], go.double())
    got = msda.ms_deform_attn_backward(*gpu(args), go.cuda(), 64)
    tol = dict(rtol=1e-8, atol=1e-9) if dt == torch.float64 else dict(rtol=2e-3, atol=2e-3)
    for g, r in zip(got, ref):
        assert g.shape == r.shape
        assert torch.allclose(g.cpu().double(), r, **tol)


@pytest.mark.parametrize("channels", [30, 32, 64, 71, 1025, 2048, 3096])   # the reference's own list (models/ops/test.py:81-86)
def test_gradcheck_like_reference(msda, channels):
    """models/ops/test.py:63-86: analytic gradient of the op vs numerical, in double."""
    from models.ops.functions import MSDeformAttnFunction
    N, M, Lq, L, P = 1, 2, 2, 2, 2
    shapes = torch.as_tensor([(6, 4), (3, 2)], dtype=torch.long).cuda()
    lsi = lsi_of(shapes)
    S = 30
    g = torch.Generator().manual_seed(3)
    value = (torch.rand(N, S, M, channels, generator=g) * 0.01).double().cuda().requires_grad_(True)
    loc = torch.rand(N, Lq, M, L, P, 2, generator=g).double().cuda().requires_grad_(True)
    aw = torch.rand(N, Lq, M, L, P, generator=g) + 1e-5
    aw = (aw / aw.sum(-1, keepdim=True).sum(-2, keepdim=True)).double().cuda().requires_grad_(True)
    assert torch.autograd.gradcheck(MSDeformAttnFunction.apply, (value, shapes, lsi, loc, aw, 2))


def test_flat_indexing_quirk_full_shape(msda, oracle):
    """SURVEY 0.6 at the real temporal-decoder shape: loc [1,300,8,R,4,2], one level."""
    R = 7
    g = torch.Generator().manual_seed(9)
    shapes = torch.as_tensor([(50, 84)], dtype=torch.long)
    value = torch.randn(1, 4200, 8, 32, generator=g)
    loc = torch.rand(1, 300, 8, R, 4, 2, generator=g)
    aw = torch.softmax(torch.randn(1, 300, 8, 4, generator=g), -1).view(1, 300, 8, 1, 4)
    ref = oracle.msda_forward(value, shapes, lsi_of(shapes), loc, aw)
    out = msda.ms_deform_attn_forward(*gpu([value, shapes, lsi_of(shapes), loc, aw]), 64).cpu()
    assert torch.allclose(out, ref, **F32_TOL)


def test_empty_and_degenerate_inputs(msda):
    shapes = torch.as_tensor([(3, 3)], dtype=torch.long).cuda()
    lsi = lsi_of(shapes)
    out = msda.ms_deform_attn_forward(torch.randn(2, 9, 8, 32).cuda(), shapes, lsi,
                                      torch.rand(2, 0, 8, 1, 4, 2).cuda(), torch.rand(2, 0, 8, 1, 4).cuda(), 64)
    assert out.shape == (2, 0, 256)
    # all samples outside the map -> exact zeros
    loc = torch.full((1, 6, 8, 1, 4, 2), 5.0).cuda()
    out = msda.ms_deform_attn_forward(torch.randn(1, 9, 8, 32).cuda(), shapes, lsi, loc,
                                      torch.rand(1, 6, 8, 1, 4).cuda(), 64)
    assert torch.count_nonzero(out) == 0
    # NaN locations are skipped, not propagated (comparison-based skip rule, cuh:288)
    loc = torch.full((1, 6, 8, 1, 4, 2), float("nan")).cuda()
    out = msda.ms_deform_attn_forward(torch.randn(1, 9, 8, 32).cuda(), shapes, lsi, loc,
                                      torch.rand(1, 6, 8, 1, 4).cuda(), 64)
    assert torch.count_nonzero(out) == 0


def test_argument_errors_mirror_the_reference(msda):
    shapes = torch.as_tensor([(3, 3)], dtype=torch.long).cuda()
    lsi = lsi_of(shapes)
    v = torch.randn(3, 9, 8, 32).cuda()
    loc = torch.rand(3, 4, 8, 1, 4, 2).cuda()
    aw = torch.rand(3, 4, 8, 1, 4).cuda()
    with pytest.raises(RuntimeError, match="contiguous"):
        msda.ms_deform_attn_forward(v.transpose(0, 1).contiguous().transpose(0, 1), shapes, lsi, loc, aw, 64)
    with pytest.raises(RuntimeError, match="im2col_step"):
        msda.ms_deform_attn_forward(v, shapes, lsi, loc, aw, 2)      # 3 % 2 != 0 (cu:50-52)
    with pytest.raises(RuntimeError, match="CUDA"):
        msda.ms_deform_attn_forward(v, shapes.cpu(), lsi, loc, aw, 64)
    with pytest.raises(RuntimeError):
        msda.ms_deform_attn_forward(v.half(), shapes, lsi, loc.half(), aw.half(), 64)


def test_runs_on_the_callers_stream_without_sync(msda, oracle):
    args = rand_case(7, 1, 8, 32, 64, 4, [(8, 8)])
    s = torch.cuda.Stream()
    dargs = gpu(args)
    torch.cuda.synchronize()
    with torch.cuda.stream(s):
        out = msda.ms_deform_attn_forward(*dargs, 64)
    s.synchronize()
    assert torch.allclose(out.cpu(), oracle.msda_forward(*args), **F32_TOL)


# ---- full-size, size-independent properties (BASELINE.json sizes; the oracle also runs) ------
def test_full_size_encoder_call(msda, oracle):
    """N=2 frames at the production geometry: S = Lq = 50*84 = 4200, L=1, M=8, D=32, P=4."""
    args = rand_case(42, 2, 8, 32, 4200, 4, [(50, 84)], lo=-0.05, hi=1.05)
    out = msda.ms_deform_attn_forward(*gpu(args), 64)
    ref = oracle.msda_forward(*args)
    assert (out.cpu() - ref).abs().max() < 1e-3
    assert torch.allclose(out.cpu(), ref, **F32_TOL)
    v, s, l, loc, aw = gpu(args)
    # linearity in value and in the attention weights
    out2 = msda.ms_deform_attn_forward(v * 2.5, s, l, loc, aw, 64)
    assert torch.allclose(out2, out * 2.5, rtol=1e-5, atol=1e-5)
    v2 = torch.randn_like(v)
    lhs = msda.ms_deform_attn_forward(v + v2, s, l, loc, aw, 64)
    rhs = out + msda.ms_deform_attn_forward(v2, s, l, loc, aw, 64)
    assert torch.allclose(lhs, rhs, rtol=1e-4, atol=1e-4)
    # a constant value map sampled strictly inside returns the constant (weights sum to 1)
    inner = loc.clamp(0.02, 0.98)
    ones = msda.ms_deform_attn_forward(torch.ones_like(v), s, l, inner, aw, 64)
    assert torch.allclose(ones, torch.ones_like(ones), atol=1e-5)
    # batch elements are independent: frame 1 alone gives the same rows
    solo = msda.ms_deform_attn_forward(v[1:].contiguous(), s, l, loc[1:].contiguous(), aw[1:].contiguous(), 64)
    assert torch.equal(solo, out[1:])
    # deterministic
    assert torch.equal(out, msda.ms_deform_attn_forward(v, s, l, loc, aw, 64))


def test_full_size_multiscale_call(msda, oracle):
    """L=4 pyramid of an 800x1333 input without dilation: 100x167, 50x84, 25x42, 13x21 = 22223 tokens."""
    shp = [(100, 167), (50, 84), (25, 42), (13, 21)]
    args = rand_case(43, 1, 8, 32, 22223, 4, shp)
    out = msda.ms_deform_attn_forward(*gpu(args), 64).cpu()
    ref = oracle.msda_forward(*args)
    assert (out - ref).abs().max() < 1e-3
    assert torch.allclose(out, ref, **F32_TOL)


# ---- fused front end ---------------------------------------------------------------------------
@pytest.mark.parametrize("L,ref_dim,Lq,shp", [
    (1, 2, 4200, [(50, 84)]), (1, 4, 300, [(50, 84)]), (4, 2, 500, [(20, 30), (10, 15), (5, 8), (3, 4)]),
    (4, 4, 300, [(20, 30), (10, 15), (5, 8), (3, 4)]), (2, 2, 99, [(7, 9), (4, 5)]), (3, 4, 64, [(7, 9), (4, 5), (2, 3)]),
])
def test_fused_front_end_matches_unfused(msda, oracle, L, ref_dim, Lq, shp):
    from dfx import ops
    g = torch.Generator().manual_seed(31 + L + ref_dim)
    N, M, D, P = 2, 8, 32, 4
    shapes = torch.as_tensor(shp, dtype=torch.long)
    S = int(shapes.prod(1).sum())
    value = torch.randn(N, S, M, D, generator=g)
    qproj = torch.randn(N, Lq, 3 * M * L * P, generator=g) * 2.0
    ref = torch.rand(N, Lq, L, ref_dim, generator=g)
    if ref_dim == 4:
        ref[..., 2:] *= 0.3
    off = qproj[..., : 2 * M * L * P].reshape(N, Lq, M, L, P, 2)
    aw = torch.softmax(qproj[..., 2 * M * L * P:].reshape(N, Lq, M, L * P), -1).view(N, Lq, M, L, P)
    if ref_dim == 2:
        norm = torch.stack([shapes[..., 1], shapes[..., 0]], -1)
        loc = ref[:, :, None, :, None, :] + off / norm[None, None, None, :, None, :]
    else:
        loc = ref[:, :, None, :, None, :2] + off / P * ref[:, :, None, :, None, 2:] * 0.5
    expect = oracle.msda_forward(value, shapes, lsi_of(shapes), loc.contiguous(), aw)
    got = ops.msda_fused_forward(value.cuda(), shapes.cuda(), lsi_of(shapes).cuda(), ref.cuda(), qproj.cuda(), L, P)
    assert torch.allclose(got.cpu(), expect, rtol=1e-4, atol=5e-5)


def test_fused_front_end_temporal_quirk(msda, oracle):
    """Lr = R reference levels against a 1-level value map: the module builds [1,Lq,8,R,4,2]
    locations and the op reads them flat."""
    from dfx import ops
    g = torch.Generator().manual_seed(77)
    R, Lq, M, D, P = 5, 300, 8, 32, 4
    shapes = torch.as_tensor([(50, 84)], dtype=torch.long)
    value = torch.randn(1, 4200, M, D, generator=g)
    qproj = torch.randn(1, Lq, 3 * M * P, generator=g)
    ref = torch.rand(1, Lq, R, 4, generator=g)
    ref[..., 2:] *= 0.3
    off = qproj[..., : 2 * M * P].reshape(1, Lq, M, 1, P, 2)
    aw = torch.softmax(qproj[..., 2 * M * P:].reshape(1, Lq, M, P), -1).view(1, Lq, M, 1, P)
    loc = (ref[:, :, None, :, None, :2] + off / P * ref[:, :, None, :, None, 2:] * 0.5).contiguous()
    assert loc.shape == (1, Lq, M, R, P, 2)
    expect = oracle.msda_forward(value, shapes, lsi_of(shapes), loc, aw)
    got = ops.msda_fused_forward(value.cuda(), shapes.cuda(), lsi_of(shapes).cuda(), ref.cuda(), qproj.cuda(), 1, P)
    assert torch.allclose(got.cpu(), expect, rtol=1e-4, atol=5e-5)


# ---- level-in-LDS kernel for single-level attention (csrc/msda_level.hip) ------------------------
def _level_case(seed, H, W, N, Lq, ref_dim, spread, raster):
    g = torch.Generator().manual_seed(seed)
    M, P, S = 8, 4, H * W
    value = torch.randn(N, S, M, 32, generator=g)
    qproj = torch.randn(N, Lq, 3 * M * P, generator=g)
    qproj[..., : 2 * M * P] *= spread
    if raster:      # encoder: the queries are the pixels of the map
        ys, xs = torch.meshgrid((torch.arange(H) + 0.5) / H, (torch.arange(W) + 0.5) / W, indexing="ij")
        xy = torch.stack([xs.reshape(-1), ys.reshape(-1)], -1).view(1, S, 1, 2).expand(N, S, 1, 2)
    else:
        xy = torch.rand(N, Lq, 1, 2, generator=g) * 1.2 - 0.1
    ref = xy.contiguous() if ref_dim == 2 else torch.cat([xy, torch.rand(N, Lq, 1, 2, generator=g) * 0.5], -1).contiguous()
    return value, qproj, ref


def _level_expect(oracle, value, qproj, ref, H, W):
    N, Lq = qproj.shape[:2]
    M, P = 8, 4
    off = qproj[..., : 2 * M * P].reshape(N, Lq, M, 1, P, 2)
    if ref.shape[-1] == 2:
        loc = ref[:, :, None, :, None, :] + off / torch.tensor([W, H], dtype=torch.float32)
    else:
        loc = ref[:, :, None, :, None, :2] + off / P * ref[:, :, None, :, None, 2:] * 0.5
    aw = torch.softmax(qproj[..., 2 * M * P:].reshape(N, Lq, M, P), -1).view(N, Lq, M, 1, P)
    shapes = torch.as_tensor([(H, W)], dtype=torch.long)
    return oracle.msda_forward(value, shapes, lsi_of(shapes), loc.contiguous(), aw)


def _run_level(value, qproj, ref, H, W, level):
    from dfx import ops
    from models.transformer_layers import make_level_tensors
    shapes, lsi = make_level_tensors([(H, W)], "cuda")
    saved = ops.USE_LEVEL_KERNEL, ops.LEVEL_MIN_QUERIES, ops.LEVEL_ON_REFERENCE_LAYOUTS
    try:
        ops.USE_LEVEL_KERNEL, ops.LEVEL_MIN_QUERIES, ops.LEVEL_ON_REFERENCE_LAYOUTS = level, 0, True
        return ops.msda_fused_forward(value.cuda(), shapes, lsi, ref.cuda(), qproj.cuda(), 1, 4)
    finally:
        ops.USE_LEVEL_KERNEL, ops.LEVEL_MIN_QUERIES, ops.LEVEL_ON_REFERENCE_LAYOUTS = saved


def _run_level_blocked(value, qproj, ref, H, W):
    """Operands re-laid out as dfx_gemm_f32's block-major outputs; the result back in [N,Lq,256]."""
    from dfx import ops
    N, S = value.shape[:2]
    Lq = qproj.shape[1]
    vb = value.view(N, S, 64, 4).permute(2, 0, 1, 3).reshape(64, N * S, 4).contiguous().cuda()
    qb = torch.cat([qproj[..., :64].view(N, Lq, 8, 8), qproj[..., 64:].view(N, Lq, 8, 4)], -1) \
        .permute(2, 0, 1, 3).reshape(8, N * Lq, 12).contiguous().cuda()
    out = ops.msda_level_forward(vb, ref.cuda(), qb, N, H, W)
    return out.view(64, N, Lq, 4).permute(1, 2, 0, 3).reshape(N, Lq, 256)


@pytest.mark.parametrize("H,W,N,Lq,ref_dim,spread,raster", [
    (50, 84, 2, 4200, 2, 3.0, True),      # production map, offsets of a few pixels
    (50, 84, 8, 4200, 2, 1.0, True),      # the bench launch: 256 workgroups, one per CU
    (50, 84, 1, 4200, 2, 30.0, True),     # large offsets, N = 1: queries split over 8 workgroups per octet
    (50, 84, 3, 300, 4, 8.0, False),      # decoder-like: few queries, box references
    (50, 84, 1, 5000, 2, 5.0, False),     # Lq != S, random references incl. outside [0,1]
    (13, 21, 3, 273, 2, 4.0, True), (7, 9, 2, 100, 4, 2.0, False), (1, 1, 2, 5, 2, 1.0, False),
    (33, 5, 1, 1111, 2, 6.0, False), (55, 86, 1, 2000, 2, 4.0, False),     # 55 x 86: about the largest map that fits
])
def test_level_kernel_matches_oracle(msda, oracle, H, W, N, Lq, ref_dim, spread, raster):
    from dfx import _lib
    assert _lib.load().dfx_msda_fused_level_fits(H, W) == 1
    value, qproj, ref = _level_case(H * 131 + W + N, H, W, N, Lq, ref_dim, spread, raster)
    expect = _level_expect(oracle, value, qproj, ref, H, W)
    got = _run_level(value, qproj, ref, H, W, True)
    plain = _run_level(value, qproj, ref, H, W, False)     # the wave-per-query kernel on the same inputs
    assert torch.allclose(got.cpu(), expect, rtol=1e-4, atol=5e-5)
    assert torch.allclose(got, plain, rtol=1e-5, atol=1e-5)
    # the block-major operand layouts the model path uses: same kernel, same arithmetic -> same bits
    assert torch.equal(_run_level_blocked(value, qproj, ref, H, W), got)


@pytest.mark.parametrize("H,W,N,Lq,ref_dim,spread", [(50, 84, 8, 4200, 2, 3.0), (50, 84, 33, 4200, 2, 2.0), (50, 84, 1, 4200, 2, 30.0),
                                                     (13, 21, 3, 273, 4, 4.0), (55, 86, 1, 2000, 2, 4.0)])
def test_level_kernel_prefetch_variant(msda, oracle, dfx_env, H, W, N, Lq, ref_dim, spread):
    """DFX_LEVEL_VARIANT=1 (csrc/msda_level.hip: 512-thread workgroups, the next item's level prefetched into registers slice by
    slice under the gather; measured slower, profiles/r04_level_variant.txt, kept for A/B runs): the same bits as the default
    schedule - per query the arithmetic is the same - also when a workgroup walks several items (N = 33: 1056 items on 256 CUs)."""
    value, qproj, ref = _level_case(H * 17 + W + N, H, W, N, Lq, ref_dim, spread, Lq == H * W)
    want = _run_level_blocked(value, qproj, ref, H, W)
    dfx_env("DFX_LEVEL_VARIANT", "1")
    got = _run_level_blocked(value, qproj, ref, H, W)
    dfx_env("DFX_LEVEL_VARIANT", None)
    assert torch.equal(got, want)


def test_level_kernel_edges(msda, oracle):
    """Samples exactly on the skip-rule bounds (h_im = -1, H-1, H; w_im likewise), NaN / inf offsets and
    -inf logits: the same answer as the oracle (a dropped sample contributes 0)."""
    H, W, N, Lq = 6, 10, 1, 64
    value, qproj, ref = _level_case(5, H, W, N, Lq, 2, 1.0, False)
    off = qproj[..., :64].view(N, Lq, 8, 4, 2)
    # location = ref + off / (W, H); pick ref = 0 and offsets that give pixel coordinates k - 0.5 exactly
    ref.zero_()
    ks = torch.tensor([-0.5, 0.0, 0.5, H - 0.5, H, H + 0.5, W - 0.5, W, W + 0.5, 1.0, 2.5, -3.0])
    for q in range(Lq):
        for m in range(8):
            for p in range(4):
                off[0, q, m, p, 0] = ks[(q + m + p) % len(ks)]           # x offset in pixels (divided by W)
                off[0, q, m, p, 1] = ks[(q * 3 + m * 5 + p * 7) % len(ks)]
    off[0, 3, 1, 2, 0] = float("nan")
    off[0, 4, 2, 1, 1] = float("inf")
    off[0, 5, 3, 0, 0] = float("-inf")
    qproj[0, 6, 64 + 4 * 2 + 1] = float("-inf")        # one masked logit of (query 6, head 2)
    expect = _level_expect(oracle, value, qproj, ref, H, W)
    got = _run_level(value, qproj, ref, H, W, True)
    assert torch.isfinite(expect).all() and torch.isfinite(got).all()
    assert torch.allclose(got.cpu(), expect, rtol=1e-4, atol=5e-5)


def test_level_kernel_refuses_large_maps(msda):
    from dfx import _lib
    lib = _lib.load()
    assert lib.dfx_msda_fused_level_fits(100, 167) == 0
    assert lib.dfx_msda_fused_level_fits(0, 5) == 0
    import ctypes
    v = torch.zeros(1, 100 * 167, 8, 32, device="cuda")
    q = torch.zeros(1, 10, 96, device="cuda")
    r = torch.zeros(1, 10, 1, 2, device="cuda")
    o = torch.zeros(1, 10, 256, device="cuda")
    ly = _lib.LevelLayout(100 * 167 * 256, 256, 32, 8, 4, 96, 8, 96, 4, 256, 32, 8, 4)
    rc = lib.dfx_msda_fused_level_forward_f32(v.data_ptr(), r.data_ptr(), 2, q.data_ptr(), q.data_ptr() + 256,
                                              ctypes.byref(ly), 1, 100, 167, 10, o.data_ptr(), None)
    assert rc < 0 and b"does not fit" in lib.dfx_last_error()
    ly.off_row = 6                                       # not a multiple of 4 floats
    rc = lib.dfx_msda_fused_level_forward_f32(v.data_ptr(), r.data_ptr(), 2, q.data_ptr(), q.data_ptr() + 256,
                                              ctypes.byref(ly), 1, 10, 10, 10, o.data_ptr(), None)
    assert rc < 0 and b"strides" in lib.dfx_last_error()
