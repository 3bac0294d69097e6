"""CPU: the C oracle against the vectors generated from the reference (tools/gen_golden_op.py)."""
import os

import numpy as np
import pytest
import torch

CASES_F32 = ["testpy_f32", "enc_l1", "dec_l1", "ms_l4", "border", "flatquirk"]
CASES_F64 = ["testpy_f64", "odd", "enc_l1_f64"]
GRAD_CASES = ["ms_l4", "border", "odd", "enc_l1_f64"]


@pytest.fixture(scope="module")
def z(golden_dir):
    return np.load(os.path.join(golden_dir, "msda_op.npz"))


def _t(z, case, key):
    return torch.from_numpy(z[f"{case}.{key}"])


@pytest.mark.parametrize("case", CASES_F32 + CASES_F64)
def test_oracle_forward_matches_reference(z, oracle, case):
    out = oracle.msda_forward(*[_t(z, case, k) for k in ("value", "shapes", "lsi", "loc", "aw")])
    ref = _t(z, case, "out")
    assert out.shape == ref.shape and out.dtype == ref.dtype
    if out.dtype == torch.float64:
        # tolerance of the reference's own double check, models/ops/test.py:39 (allclose defaults)
        assert torch.allclose(out, ref, rtol=1e-5, atol=1e-8)
    else:
        # the reference's float check uses rtol 1e-2 / atol 1e-3 (test.py:55); we hold 1e-5
        assert torch.allclose(out, ref, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("case", GRAD_CASES)
def test_oracle_backward_matches_reference_autograd(z, oracle, case):
    args = [_t(z, case, k) for k in ("value", "shapes", "lsi", "loc", "aw", "grad_out")]
    args = [a.double() if a.is_floating_point() else a for a in args]
    gv, gl, ga = oracle.msda_backward(*args)
    assert torch.allclose(gv, _t(z, case, "grad_value"), rtol=1e-9, atol=1e-11)
    assert torch.allclose(gl, _t(z, case, "grad_loc"), rtol=1e-9, atol=1e-10)
    assert torch.allclose(ga, _t(z, case, "grad_aw"), rtol=1e-9, atol=1e-11)


def test_oracle_flat_read_uses_prefix_only(z, oracle):
    """a location tensor R times too large is read as a packed prefix."""
    v, s, l, loc, aw = [_t(z, "flatquirk", k) for k in ("value", "shapes", "lsi", "loc", "aw")]
    N, Lq, M, R, P, _ = loc.shape
    prefix = loc.reshape(-1)[: N * Lq * M * P * 2].view(N, Lq, M, 1, P, 2).contiguous()
    assert torch.equal(oracle.msda_forward(v, s, l, loc, aw), oracle.msda_forward(v, s, l, prefix, aw))


def test_oracle_threads_do_not_change_results(z, oracle):
    args = [_t(z, "enc_l1", k) for k in ("value", "shapes", "lsi", "loc", "aw")]
    oracle.set_threads(1)
    a = oracle.msda_forward(*args)
    oracle.set_threads(4)
    b = oracle.msda_forward(*args)
    assert torch.equal(a, b)


def test_core_pytorch_mirror_matches_reference(z):
    """models.ops.functions.ms_deform_attn_core_pytorch (API mirror, debug aid)."""
    from models.ops.functions import ms_deform_attn_core_pytorch
    for case in ("testpy_f64", "odd", "ms_l4", "border"):
        v, s, loc, aw = [_t(z, case, k) for k in ("value", "shapes", "loc", "aw")]
        out = ms_deform_attn_core_pytorch(v, s, loc, aw)
        tol = dict(rtol=1e-5, atol=1e-8) if v.dtype == torch.float64 else dict(rtol=1e-4, atol=1e-5)
        assert torch.allclose(out, _t(z, case, "out"), **tol), case


def test_oracle_roi_align_simple(oracle):
    # constant map -> every bin equals the constant; box fully inside
    x = torch.full((1, 2, 8, 8), 3.0)
    rois = torch.tensor([[0, 1.0, 1.0, 6.0, 6.0]])
    out = oracle.roi_align(x, rois, 7, 1.0, 2, True)
    assert torch.allclose(out, torch.full_like(out, 3.0))
    # linear ramp in x: bilinear sampling of a linear function is exact away from borders
    ramp = torch.arange(8.0).view(1, 1, 1, 8).expand(1, 1, 8, 8).contiguous()
    out = oracle.roi_align(ramp, torch.tensor([[0, 2.0, 2.0, 6.0, 6.0]]), 2, 1.0, 2, True)
    # aligned: x1 = 2 - 0.5 = 1.5, bin width 2 -> bin centres at 2.5 and 4.5
    assert torch.allclose(out[0, 0, 0], torch.tensor([2.5, 4.5]))
