"""GPU: the hand-written convolutions (csrc/conv_igemm.hip, csrc/conv_wino.hip) against float64 conv2d.
The products run on exact-fp32 MFMA; Winograd F(2x2,3x3) adds the rounding of its transforms (inputs
up to 4x, filters exact halves/quarters), so its bound is a few times the direct form's."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _ref(x, w, b, stride, pad, dil, act):
    y = F.conv2d(x.double(), w.double(), None if b is None else b.double(), stride, pad, dil)
    if act == "relu":
        y = y.relu()
    elif act == "gelu":
        y = F.gelu(y)
    return y


CASES = [  # N, Ci, Co, H, W, k, stride, pad, dil, act
    (2, 3, 64, 67, 90, 7, 2, 3, 1, None),          # ResNet stem geometry
    (2, 64, 64, 25, 41, 3, 1, 1, 1, "relu"),       # layer1 conv2
    (2, 128, 128, 27, 35, 3, 2, 1, 1, "relu"),     # layer2.0 conv2 (stride 2, odd map)
    (1, 256, 256, 20, 34, 3, 2, 1, 1, "relu"),     # layer3.0 conv2
    (2, 64, 128, 13, 21, 3, 1, 2, 2, "relu"),      # dilated
    (3, 1, 16, 40, 67, 3, 2, 1, 1, "gelu"),        # DFormer stem, first convolution
    (2, 16, 32, 20, 34, 3, 2, 1, 1, None),
    (2, 32, 64, 21, 33, 3, 2, 1, 1, None),
    (1, 24, 40, 9, 11, 5, 1, 2, 1, "relu"),        # odd everything
    (1, 8, 200, 6, 7, 1, 1, 0, 1, None),           # 1x1, Co not a multiple of the tile
]


@pytest.mark.parametrize("N,Ci,Co,H,W,k,stride,pad,dil,act", CASES)
def test_igemm_matches_fp64(N, Ci, Co, H, W, k, stride, pad, dil, act):
    from dfx import ops
    g = torch.Generator().manual_seed(Ci * 131 + Co + H)
    x = torch.randn(N, Ci, H, W, generator=g).cuda()
    w = (torch.randn(Co, Ci, k, k, generator=g) / (Ci * k * k) ** 0.5).cuda()
    b = torch.randn(Co, generator=g).cuda()
    plan = ops.ConvPlan(w, b, stride, pad, dil, act, algo="igemm")
    got = plan(x)
    want = _ref(x, w, b, stride, pad, dil, act)
    assert got.shape == want.shape
    assert (got.double() - want).abs().max().item() < 4e-6 * (Ci * k * k) ** 0.5


TILE = [  # N, Ci, Co, H, W, k, stride, pad, act   (csrc/conv_tile.hip: few input channels, the input tile lives in LDS)
    (2, 3, 64, 67, 90, 7, 2, 3, None),             # ResNet stem geometry, odd map: partial tiles on both axes
    (3, 3, 64, 96, 128, 7, 2, 3, "relu"),          # several tiles per image, image boundaries inside the batch
    (3, 1, 16, 40, 67, 3, 2, 1, "gelu"),           # DFormer stem, first convolution (one K-step, 16 of 32 channel rows)
    (2, 1, 16, 33, 31, 3, 2, 1, None),
    (2, 2, 24, 19, 37, 3, 1, 1, "relu"),           # stride 1 (no parity split of the staged rows), Co not a multiple of 8
    (1, 3, 40, 50, 21, 5, 2, 2, None),             # 5x5/2, K = 75 -> 80, two 32-channel blocks with 24 idle rows
    (5, 3, 7, 9, 8, 3, 1, 0, None),                # no padding: the output is smaller than the input; tiny maps
]


@pytest.mark.parametrize("N,Ci,Co,H,W,k,stride,pad,act", TILE)
def test_tile_kernel_matches_fp64_and_the_implicit_gemm(N, Ci, Co, H, W, k, stride, pad, act):
    from dfx import ops
    g = torch.Generator().manual_seed(Ci * 17 + Co + H + k)
    x = torch.randn(N, Ci, H, W, generator=g).cuda()
    w = (torch.randn(Co, Ci, k, k, generator=g) / (Ci * k * k) ** 0.5).cuda()
    b = torch.randn(Co, generator=g).cuda()
    plan = ops.ConvPlan(w, b, stride, pad, 1, act)
    assert plan.algo == "tile"
    got = plan(x)
    want = _ref(x, w, b, stride, pad, 1, act)
    assert got.shape == want.shape
    assert (got.double() - want).abs().max().item() < 4e-6 * (Ci * k * k) ** 0.5
    other = ops.ConvPlan(w, b, stride, pad, 1, act, algo="igemm")(x)
    assert (got - other).abs().max().item() < 4e-6 * (Ci * k * k) ** 0.5
    for _ in range(3):                                       # run to run: same bits (double-buffered staging, persistent walk)
        assert torch.equal(plan(x), got)
    # without a bias
    assert (ops.ConvPlan(w, None, stride, pad, 1, None)(x).double() - _ref(x, w, None, stride, pad, 1, None)).abs().max().item() < 4e-6 * (Ci * k * k) ** 0.5


def test_tile_kernel_reads_channel_planes_of_a_wider_clip_in_place():
    """The stems read the RGB planes / the depth plane of the [T,4,H,W] clip in place (image stride argument)."""
    from dfx import ops
    g = torch.Generator().manual_seed(5)
    clip = torch.randn(3, 4, 70, 101, generator=g).cuda()
    w3 = (torch.randn(64, 3, 7, 7, generator=g) / 147 ** 0.5).cuda()
    w1 = (torch.randn(16, 1, 3, 3, generator=g) / 3.0).cuda()
    for w, sl, st, pd in ((w3, slice(0, 3), 2, 3), (w1, slice(3, 4), 2, 1)):
        plan = ops.ConvPlan(w, None, st, pd, 1, "relu")
        assert plan.algo == "tile"
        got = plan(clip[:, sl])
        want = _ref(clip[:, sl].contiguous(), w, None, st, pd, 1, "relu")
        assert (got.double() - want).abs().max().item() < 6e-5


def test_tile_kernel_at_production_size_against_the_implicit_gemm():
    """800 x 1333 stem and depth-stem convolutions of a 4-frame block: every tile position incl. the ragged right / bottom
    edges, a persistent walk of 8400 tiles over 512 workgroups."""
    from dfx import ops
    g = torch.Generator().manual_seed(6)
    clip = torch.randn(4, 4, 800, 1333, generator=g).cuda()
    for Co, sl, k, pd in ((64, slice(0, 3), 7, 3), (16, slice(3, 4), 3, 1)):
        Ci = sl.stop - sl.start
        w = (torch.randn(Co, Ci, k, k, generator=g) / (Ci * k * k) ** 0.5).cuda()
        b = torch.randn(Co, generator=g).cuda()
        got = ops.ConvPlan(w, b, 2, pd, 1, "relu")(clip[:, sl])
        ref = ops.ConvPlan(w, b, 2, pd, 1, "relu", algo="igemm")(clip[:, sl])
        assert got.shape == ref.shape == (4, Co, 400, 667)
        assert (got - ref).abs().max().item() < 4e-6 * (Ci * k * k) ** 0.5


WINO = [  # N, Ci, Co, H, W, dil
    (2, 64, 64, 24, 40, 1),
    (3, 64, 64, 25, 41, 1),        # odd map: edge tiles with one valid row / column
    (1, 128, 128, 50, 84, 1),
    (2, 256, 256, 13, 21, 1),
    (2, 512, 512, 13, 21, 2),      # DC5 stage
    (1, 64, 128, 50, 84, 2),
    (5, 8, 64, 7, 5, 1),           # tiles of several images inside one workgroup
    (2, 16, 64, 9, 10, 3),
    (8, 64, 64, 100, 167, 1),      # 525 logical blocks: two full rounds of 64x64 workgroups + 13 blocks as quarter-size ones
    (4, 256, 256, 50, 84, 2),      # 276 blocks
]


@pytest.mark.parametrize("N,Ci,Co,H,W,dil", WINO)
@pytest.mark.parametrize("act", [None, "relu"])
def test_winograd_matches_fp64(N, Ci, Co, H, W, dil, act):
    from dfx import ops
    g = torch.Generator().manual_seed(Ci * 7 + Co + H + dil)
    x = torch.randn(N, Ci, H, W, generator=g).cuda()
    w = (torch.randn(Co, Ci, 3, 3, generator=g) / (Ci * 9) ** 0.5).cuda()
    b = torch.randn(Co, generator=g).cuda()
    scale = (torch.rand(Co, generator=g) + 0.5).cuda()
    plan = ops.ConvPlan(w, b, 1, dil, dil, act, scale=scale)
    assert plan.algo == "wino"
    got = plan(x)
    want = _ref(x, w * scale.view(-1, 1, 1, 1), b, 1, dil, dil, act)
    assert got.shape == want.shape
    assert (got.double() - want).abs().max().item() < 1.5e-5 * (Ci * 9) ** 0.5 / 3


def test_winograd_equals_igemm_closely_on_inf_free_borders():
    """The two algorithms on one problem (validity masks of edge tiles, dilation phases)."""
    from dfx import ops
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 64, 50, 84, generator=g).cuda()
    w = (torch.randn(64, 64, 3, 3, generator=g) / 24).cuda()
    for dil in (1, 2):
        a = ops.ConvPlan(w, None, 1, dil, dil, None, algo="wino")(x)
        b = ops.ConvPlan(w, None, 1, dil, dil, None, algo="igemm")(x)
        assert (a - b).abs().max().item() < 2e-5


def test_conv_plan_rejects_cpu_and_bad_geometry():
    from dfx import ops
    with pytest.raises(RuntimeError):
        ops.ConvPlan(torch.randn(64, 64, 3, 3), None, 1, 1, 1)
    w = torch.randn(64, 60, 3, 3).cuda()
    assert ops.ConvPlan(w, None, 1, 1, 1).algo == "igemm"       # Ci % 8 != 0 -> implicit GEMM
    with pytest.raises(RuntimeError):
        ops.ConvPlan(w, None, 1, 1, 1, algo="wino")


def test_igemm_reads_a_channel_slice_in_place():
    """x[:, :3] / x[:, 3:4] of an RGB-D clip go to the stem convolutions without a copy (image stride argument)."""
    from dfx import ops
    g = torch.Generator().manual_seed(9)
    x = torch.randn(3, 4, 37, 50, generator=g).cuda()
    for sl, ci in ((slice(0, 3), 3), (slice(3, 4), 1)):
        w = torch.randn(16, ci, 3, 3, generator=g).cuda()
        plan = ops.ConvPlan(w, None, 2, 1, 1, None)
        assert torch.equal(plan(x[:, sl]), plan(x[:, sl].contiguous()))


@pytest.mark.parametrize("Ci,Co,H,W,dil", [(64, 64, 200, 334, 1), (128, 128, 100, 167, 1), (512, 512, 50, 84, 2)])
def test_convolutions_at_production_sizes_agree_and_are_linear(Ci, Co, H, W, dil):
    """The ResNet maps of an 800x1333 frame: the two algorithms against each other (different arithmetic, same
    convolution) and linearity conv(a x + y) = a conv(x) + conv(y) of each - size-independent checks where an fp64
    reference of the whole map would take minutes on the host."""
    from dfx import ops
    g = torch.Generator().manual_seed(Ci + H)
    x = torch.randn(2, Ci, H, W, generator=g).cuda()
    y = torch.randn(2, Ci, H, W, generator=g).cuda()
    w = (torch.randn(Co, Ci, 3, 3, generator=g) / (Ci * 9) ** 0.5).cuda()
    wino = ops.ConvPlan(w, None, 1, dil, dil, None, algo="wino")
    igemm = ops.ConvPlan(w, None, 1, dil, dil, None, algo="igemm")
    a, b = wino(x), igemm(x)
    assert (a - b).abs().max().item() < 3e-5
    for plan in (wino, igemm):
        lhs = plan(2.5 * x + y)
        rhs = 2.5 * plan(x) + plan(y)
        assert (lhs - rhs).abs().max().item() < 2e-4          # outputs of magnitude ~10, three roundings of K = 9 Ci sums
    # a one-hot input reproduces the (flipped) filter taps around the pixel: exact placement of every tap, incl. dilation
    e = torch.zeros(1, Ci, H, W).cuda()
    e[0, 3, H // 2, W // 2] = 1.0
    out = wino(e)
    for ky in range(3):
        for kx in range(3):
            got = out[0, :, H // 2 - (ky - 1) * dil, W // 2 - (kx - 1) * dil]
            assert (got - w[:, 3, ky, kx]).abs().max().item() < 1e-6
