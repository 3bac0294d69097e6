"""GPU: the hand-written fp32 MFMA GEMM (csrc/gemm_f32.hip) against a float64 reference.
fp32 MFMA is an exact-fp32 k-ordered fma chain, so the error bound is the usual sum-of-products
one: we assert 4e-6 * sqrt(K) * scale."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref_linear(x, w, b, relu, res, add, mask):
    y = (x.double() + (0 if add is None else add.double())) @ w.double().t()
    if b is not None:
        y = y + b.double()
    if res is not None:
        y = y + res.double()
    if relu:
        y = y.relu()
    if mask is not None:
        y = y.masked_fill(mask[..., None], 0.0)
    return y


@pytest.mark.parametrize("M,N,K", [(4200, 256, 256), (33600, 96, 256), (300, 256, 256), (1000, 1024, 256),
                                   (777, 256, 1024), (64, 128, 2048), (130, 3, 256), (5, 4, 8), (2500, 64, 128),
                                   (129, 300, 260)])
@pytest.mark.parametrize("variant", ["plain", "bias_relu", "all"])
def test_linear_matches_fp64(M, N, K, variant):
    from dfx import ops
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g).cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    b = torch.randn(N, generator=g).cuda() if variant != "plain" else None
    res = torch.randn(M, N, generator=g).cuda() if variant == "all" else None
    add = torch.randn(M, K, generator=g).cuda() if variant == "all" else None
    mask = (torch.rand(M, generator=g) > 0.8).cuda() if variant == "all" else None
    got = ops.linear(x, w, b, relu=variant != "plain", residual=res, add=add, row_mask=mask)
    want = _ref_linear(x, w, b, variant != "plain", res, add, mask)
    tol = 4e-6 * K ** 0.5 * (2.0 if add is not None else 1.0)
    assert got.shape == (M, N)
    assert (got.double() - want).abs().max().item() < tol


@pytest.mark.parametrize("M,N,K", [(1200, 256, 256), (300, 96, 256), (2400, 512, 512), (37, 256, 1024), (4800, 32, 256),
                                   (1201, 480, 1024), (1200, 4, 256), (700, 3, 1024), (64, 2, 512), (1200, 1024, 256)])
@pytest.mark.parametrize("variant", ["plain", "bias_relu", "add_res_gelu"])
def test_linear_rows_kernel_matches_fp64_and_the_tile_kernel(M, N, K, variant, dfx_env):
    """Few rows (the 300-query layers): one 32 x 32 tile per workgroup, K split over its waves (csrc/gemm_f32.hip,
    linear_rows_kernel) - against fp64 and against the tile kernel on the same operands (DFX_GEMM_NO_ROWS=1)."""
    from dfx import ops
    g = torch.Generator().manual_seed(3 * M + N + K)
    x = torch.randn(M, K, generator=g).cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    b = torch.randn(N, generator=g).cuda() if variant != "plain" else None
    res = torch.randn(M, N, generator=g).cuda() if variant == "add_res_gelu" else None
    add = torch.randn(M, K, generator=g).cuda() if variant == "add_res_gelu" else None
    act = {"plain": None, "bias_relu": "relu", "add_res_gelu": "gelu"}[variant]
    ops.profile_start()
    got = ops.linear(x, w, b, residual=res, add=add, act=act)
    tiles = [tb for (_, _, ta, tb) in ops.profile_stop() if ta == -2]
    assert tiles == [32032], "the rows kernel serves this shape"
    dfx_env("DFX_GEMM_NO_ROWS", "1")
    other = ops.linear(x, w, b, residual=res, add=add, act=act)
    dfx_env("DFX_GEMM_NO_ROWS", None)
    y = (x.double() + (add.double() if add is not None else 0)) @ w.double().t()
    if b is not None:
        y = y + b.double()
    if res is not None:
        y = y + res.double()
    want = y.relu() if act == "relu" else torch.nn.functional.gelu(y) if act == "gelu" else y
    tol = 4e-6 * K ** 0.5 * (2.0 if add is not None else 1.0)
    assert (got.double() - want).abs().max().item() < tol
    assert (got - other).abs().max().item() < tol
    for _ in range(5):                                   # run-to-run: same bits
        assert torch.equal(ops.linear(x, w, b, residual=res, add=add, act=act), got)


@pytest.mark.parametrize("M,N,K,w", [(4200, 256, 256, 4), (8400, 96, 256, 12), (333, 96, 64, 12), (130, 256, 260, 8),
                                     (5, 12, 8, 4)])
def test_linear_block_major_layouts(M, N, K, w, dfx_env):
    """col_block stores C as [N/w][M][w]; x_blocked reads A as [K/4][M][4] - the layouts between the
    projections and the level-in-LDS MSDA kernel.  Same arithmetic as the row-major call of the tile kernel: equal bits
    (the few-row kernel, which serves row-major operands only, sums K in another order: kept out of the comparison)."""
    from dfx import ops
    dfx_env("DFX_GEMM_NO_ROWS", "1")
    g = torch.Generator().manual_seed(M * 7 + N + K + w)
    x = torch.randn(M, K, generator=g).cuda()
    wt = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    b = torch.randn(N, generator=g).cuda()
    mask = (torch.rand(M, generator=g) > 0.8).cuda()
    plain = ops.linear(x, wt, b, row_mask=mask)
    blk = ops.linear(x, wt, b, row_mask=mask, col_block=w)
    assert blk.shape == (N // w, M, w)
    assert torch.equal(blk.permute(1, 0, 2).reshape(M, N), plain)
    xb = x.view(M, K // 4, 4).permute(1, 0, 2).contiguous()
    res = torch.randn(M, N, generator=g).cuda()
    assert torch.equal(ops.linear(xb, wt, b, relu=True, residual=res, x_blocked=True),
                       ops.linear(x, wt, b, relu=True, residual=res))


@pytest.mark.parametrize("Nb,Ci,Co,H,W", [(2, 64, 256, 20, 34), (3, 256, 64, 10, 18), (1, 1024, 512, 8, 12),
                                          (2, 2048, 256, 5, 8), (2, 128, 256, 50, 84)])
def test_conv1x1_matches_torch(Nb, Ci, Co, H, W):
    from dfx import ops
    g = torch.Generator().manual_seed(Ci + Co)
    x = torch.randn(Nb, Ci, H, W, generator=g).cuda()
    w = (torch.randn(Co, Ci, 1, 1, generator=g) / Ci ** 0.5).cuda()
    b = torch.randn(Co, generator=g).cuda()
    r = torch.randn(Nb, Co, H, W, generator=g).cuda()
    want = (torch.nn.functional.conv2d(x.double(), w.double(), b.double()) + r.double()).relu()
    got = ops.conv1x1(x, w, b, residual=r, relu=True)
    assert (got.double() - want).abs().max().item() < 4e-6 * Ci ** 0.5
    if (((H + 1) // 2) * ((W + 1) // 2)) % 4 == 0:
        want2 = torch.nn.functional.conv2d(x.double(), w.double(), None, stride=2)
        got2 = ops.conv1x1(x, w, stride=2)
        assert (got2.double() - want2).abs().max().item() < 4e-6 * Ci ** 0.5


@pytest.mark.parametrize("rows,C", [(4200, 256), (33, 64), (7, 1024), (300, 512), (5, 260)])
def test_add_layernorm_matches_torch(rows, C):
    from dfx import ops
    g = torch.Generator().manual_seed(rows + C)
    x = (torch.randn(rows, C, generator=g) * 3 + 1).cuda()
    r = torch.randn(rows, C, generator=g).cuda()
    ln = torch.nn.LayerNorm(C).cuda()
    with torch.no_grad():
        ln.weight.copy_(torch.randn(C, generator=g))
        ln.bias.copy_(torch.randn(C, generator=g))
        assert torch.allclose(ops.add_layernorm(x, r, ln), ln(x + r), atol=2e-5, rtol=1e-5)
        assert torch.allclose(ops.add_layernorm(x, None, ln), ln(x), atol=2e-5, rtol=1e-5)


@pytest.mark.parametrize("shape", [(2, 64, 50, 67), (1, 3, 7, 9), (2, 5, 8, 8), (1, 2, 1, 1)])
def test_stem_epilogue_matches_torch(shape):
    from dfx import ops
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(*shape, generator=g).cuda()
    b = torch.randn(shape[1], generator=g).cuda()
    want = torch.nn.functional.max_pool2d((x + b.view(1, -1, 1, 1)).relu(), 3, 2, 1)
    assert torch.equal(ops.bias_relu_maxpool(x, b), want)


@pytest.mark.parametrize("rows,ref_dim", [(300, 4), (2400, 2), (1, 4), (777, 2)])
def test_box_refine_matches_torch(rows, ref_dim):
    """sigmoid(delta + inverse_sigmoid(ref)) in one launch vs the reference's chain of elementwise ops
    (util/misc.py inverse_sigmoid), incl. references at / beyond 0 and 1."""
    from dfx import ops
    from util.misc import inverse_sigmoid
    g = torch.Generator().manual_seed(rows + ref_dim)
    delta = (torch.randn(rows, 4, generator=g) * 2).cuda()
    ref = torch.rand(rows, ref_dim, generator=g)
    ref.view(-1)[::7] = 0.0
    ref.view(-1)[3::11] = 1.0
    ref.view(-1)[5::13] = 1.2
    ref.view(-1)[6::17] = -0.1
    ref = ref.cuda()
    want = delta.clone()
    want[..., :ref_dim] += inverse_sigmoid(ref)
    want = want.sigmoid()
    got = ops.box_refine(delta, ref)
    assert torch.allclose(got, want, rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("B,Lq,Lk,mode", [(2, 300, 300, "self_pos"), (3, 300, 2480, "cross"), (4, 100, 100, "same"),
                                          (1, 65, 33, "cross"), (2, 1, 1, "same"), (2, 64, 32, "self_pos"),
                                          (1, 31, 95, "cross"), (40, 300, 300, "same")])
@pytest.mark.parametrize("groups", [None, "1", "2", "4"])
def test_fused_mha_matches_module(B, Lq, Lk, mode, groups, dfx_env):
    """models/fused_mha.py (GEMM projections + csrc/mha.hip) against nn.MultiheadAttention in eval mode; the key range split
    over 1 / 2 / 4 wave groups of a workgroup (None: the launch's own choice)."""
    from models import fused_mha
    if groups is not None:
        dfx_env("DFX_MHA_GROUPS", groups)
    torch.manual_seed(B * 1000 + Lq + Lk)
    mod = torch.nn.MultiheadAttention(256, 8, dropout=0.1).cuda().eval()
    with torch.no_grad():
        mod.in_proj_bias.normal_(0, 0.2)
        mod.out_proj.bias.normal_(0, 0.2)
        x = torch.randn(B, Lq, 256, device="cuda") * 2
        if mode == "same":
            q = k = v = x
        elif mode == "self_pos":
            q = k = x + torch.randn(B, Lq, 256, device="cuda")
            v = x
        else:
            q = x
            k = torch.randn(B, Lk, 256, device="cuda") * 2
            v = torch.randn(B, Lk, 256, device="cuda")
        assert fused_mha.usable(mod, q, k, v)
        want = mod(q.transpose(0, 1), k.transpose(0, 1), v.transpose(0, 1))[0].transpose(0, 1)
        got = fused_mha.forward(mod, q, k, v)
    assert got.shape == want.shape
    assert torch.allclose(got, want, rtol=1e-4, atol=2e-5), (got - want).abs().max().item()


@pytest.mark.parametrize("K,R", [(600, 49), (37, 49), (1, 49), (300, 64), (5, 7)])
def test_dynamic_conv_matches_module(K, R):
    """csrc/dynconv.hip against the reference formulation of DynamicConv (two bmm + LayerNorm + ReLU)."""
    from dfx import ops
    torch.manual_seed(K + R)
    C, dd = 256, 64
    feats = torch.randn(K, R, C, device="cuda")
    params = torch.randn(K, 2 * C * dd, device="cuda") / 8
    n1, n2 = torch.nn.LayerNorm(dd).cuda(), torch.nn.LayerNorm(C).cuda()
    with torch.no_grad():
        for n in (n1, n2):
            n.weight.normal_(1, 0.3)
            n.bias.normal_(0, 0.3)
        k1 = params[:, : C * dd].reshape(K, C, dd)
        k2 = params[:, C * dd:].reshape(K, dd, C)
        want = torch.relu(n2(torch.bmm(torch.relu(n1(torch.bmm(feats, k1))), k2)))
        got = ops.dynamic_conv(feats, params, n1, n2)
    assert got.shape == want.shape
    assert torch.allclose(got, want, rtol=2e-4, atol=2e-4), (got - want).abs().max().item()


@pytest.mark.parametrize("M,N,K", [(2400, 256, 12544), (1200, 256, 12544), (300, 256, 4100), (77, 64, 2048)])
def test_linear_split_k_matches_fp64(M, N, K):
    """Few output tiles and a long K (RCNNHead.out_layer): ops.linear cuts K into ranges + one reduction launch."""
    from dfx import ops
    assert ops._split_k(M, N, K) > 1
    g = torch.Generator().manual_seed(M + K)
    x = torch.randn(M, K, generator=g).cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    b = torch.randn(N, generator=g).cuda()
    res = torch.randn(M, N, generator=g).cuda()
    got = ops.linear(x, w, b, relu=True, residual=res)
    want = _ref_linear(x, w, b, True, res, None, None)
    assert (got.double() - want).abs().max().item() < 4e-6 * K ** 0.5


def test_linear_gelu_epilogue():
    from dfx import ops
    g = torch.Generator().manual_seed(3)
    x = torch.randn(4200, 256, generator=g).cuda()
    w = (torch.randn(256, 256, generator=g) / 16).cuda()
    b = torch.randn(256, generator=g).cuda()
    want = torch.nn.functional.gelu(x.double() @ w.double().t() + b.double())
    assert (ops.linear(x, w, b, act="gelu").double() - want).abs().max().item() < 1e-5


def test_fused_linear_module_routes_to_the_gemm_and_keeps_state_dict_keys():
    from models.fused import Linear
    lin = Linear(256, 3).cuda()
    assert list(lin.state_dict()) == ["weight", "bias"]
    x = torch.randn(2, 300, 256).cuda()
    with torch.no_grad():
        got = lin(x)
    want = torch.nn.functional.linear(x.double(), lin.weight.double(), lin.bias.double())
    assert got.shape == (2, 300, 3) and (got.double() - want).abs().max().item() < 1e-5
    assert lin(x).requires_grad          # grad mode: plain nn.Linear


@pytest.mark.parametrize("N,C,H,W", [(2, 256, 50, 84), (3, 256, 13, 21), (1, 64, 7, 5)])
def test_group_norm_matches_torch(N, C, H, W):
    from dfx import ops
    g = torch.Generator().manual_seed(C + H)
    x = (torch.randn(N, C, H, W, generator=g) * 3 + 1.5).cuda()
    gn = torch.nn.GroupNorm(32, C).cuda()
    with torch.no_grad():
        gn.weight.copy_(torch.randn(C, generator=g))
        gn.bias.copy_(torch.randn(C, generator=g))
        want = torch.nn.functional.group_norm(x.double(), 32, gn.weight.double(), gn.bias.double(), gn.eps)
        a = ops.group_norm(x, gn)
        b = ops.group_norm(x, gn, tokens_out=True)
    assert (a.double() - want).abs().max().item() < 2e-5
    assert b.shape == x.shape and torch.equal(b.contiguous(), a)
    assert b.flatten(2).transpose(1, 2).is_contiguous()       # what the transformer does next is free


def test_lds_dma_staging_is_bit_identical_to_register_staging_repeatedly(dfx_env):
    """The LDS-DMA staged kernel (default) against the register-staged one (DFX_GEMM_NO_DMA=1) on the same operands,
    many launches per shape: identical bits every time - a landing-order race would show up as a stray tile."""
    from dfx import ops
    g = torch.Generator().manual_seed(77)
    for kind, shape in (("linear", (33600, 256, 1024)), ("linear", (4200, 1024, 256)), ("linear", (1200, 256, 256)),
                        ("conv", (4, 1024, 2048, 50, 84)), ("conv", (2, 256, 64, 200, 334))):
        if kind == "linear":
            M, N, K = shape
            x = torch.randn(M, K, generator=g).cuda()
            w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
            b = torch.randn(N, generator=g).cuda()
            run = lambda: ops.linear(x, w, b, relu=True)
        else:
            Nb, Ci, Co, H, W = shape
            x = torch.randn(Nb, Ci, H, W, generator=g).cuda()
            w = (torch.randn(Co, Ci, 1, 1, generator=g) / Ci ** 0.5).cuda()
            b = torch.randn(Co, generator=g).cuda()
            run = lambda: ops.conv1x1(x, w, b, relu=True)
        dfx_env("DFX_GEMM_NO_DMA", "1")
        ref = run()
        dfx_env("DFX_GEMM_NO_DMA", None)
        for _ in range(25):
            assert torch.equal(run(), ref), (kind, shape)


def test_linear_fuzz_shapes_match_fp64():
    """Random small shapes: every tile family, K with and without a 16-tail (LDS-DMA / register staging), N not a
    multiple of 4 (narrow epilogue), with / without bias, residual, prologue add and row mask."""
    from dfx import ops
    g = torch.Generator().manual_seed(2024)
    for trial in range(40):
        M = int(torch.randint(1, 700, (1,), generator=g))
        N = int(torch.randint(1, 300, (1,), generator=g))
        K = int(torch.randint(1, 137, (1,), generator=g)) * 4
        flags = torch.randint(0, 2, (5,), generator=g).tolist()
        x = torch.randn(M, K, generator=g).cuda()
        w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
        b = torch.randn(N, generator=g).cuda() if flags[0] else None
        res = torch.randn(M, N, generator=g).cuda() if flags[1] else None
        add = torch.randn(M, K, generator=g).cuda() if flags[2] else None
        mask = (torch.rand(M, generator=g) > 0.7).cuda() if flags[3] else None
        got = ops.linear(x, w, b, relu=bool(flags[4]), residual=res, add=add, row_mask=mask)
        want = _ref_linear(x, w, b, bool(flags[4]), res, add, mask)
        err = (got.double() - want).abs().max().item()
        assert err < 6e-6 * K ** 0.5 * (2.0 if add is not None else 1.0), (trial, M, N, K, flags, err)


def test_conv1x1_fuzz_shapes_match_fp64():
    from dfx import ops
    g = torch.Generator().manual_seed(7)
    for trial in range(20):
        Nb = int(torch.randint(1, 4, (1,), generator=g))
        Ci = int(torch.randint(1, 40, (1,), generator=g)) * 4
        Co = int(torch.randint(1, 200, (1,), generator=g))
        H = int(torch.randint(1, 12, (1,), generator=g)) * 2
        W = int(torch.randint(1, 12, (1,), generator=g)) * 2
        x = torch.randn(Nb, Ci, H, W, generator=g).cuda()
        w = (torch.randn(Co, Ci, 1, 1, generator=g) / Ci ** 0.5).cuda()
        b = torch.randn(Co, generator=g).cuda()
        r = torch.randn(Nb, Co, H, W, generator=g).cuda()
        want = (torch.nn.functional.conv2d(x.double(), w.double(), b.double()) + r.double()).relu()
        got = ops.conv1x1(x, w, b, residual=r, relu=True)
        assert (got.double() - want).abs().max().item() < 6e-6 * Ci ** 0.5, (trial, Nb, Ci, Co, H, W)


def test_linear_runs_in_row_ranges_when_an_operand_would_pass_2_gib(monkeypatch):
    """The kernel addresses A / the residual with 32-bit byte offsets; ops.linear hands more rows than that over in
    row ranges (every layout is row-separable).  The limit is lowered here so that small tensors take the path:
    identical bits to the one-call result, for the plain, the residual / add / mask and the column-block-major forms."""
    from dfx import ops
    g = torch.Generator().manual_seed(77)
    M, N, K = 1000, 96, 256
    x = torch.randn(M, K, generator=g).cuda()
    w = (torch.randn(N, K, generator=g) / 16).cuda()
    b = torch.randn(N, generator=g).cuda()
    res = torch.randn(M, N, generator=g).cuda()
    add = torch.randn(M, K, generator=g).cuda()
    mask = (torch.rand(M, generator=g) > 0.7).cuda()
    whole = (ops.linear(x, w, b), ops.linear(x, w, b, relu=True, residual=res, add=add, row_mask=mask),
             ops.linear(x, w, b, row_mask=mask, col_block=12))
    monkeypatch.setattr(ops, "_GEMM_MAX_BYTES", 300 * K * 4)          # -> ranges of 256 rows
    parts = (ops.linear(x, w, b), ops.linear(x, w, b, relu=True, residual=res, add=add, row_mask=mask),
             ops.linear(x, w, b, row_mask=mask, col_block=12))
    for a, c in zip(whole, parts):
        assert torch.equal(a, c)


@pytest.mark.timeout(600)
def test_linear_just_above_the_2_gib_operand_limit():
    """K = 1024 (the encoder FFN's second Linear) with 2^19 + 130 rows: A is 2 GiB + 520 KiB.  Checked on sampled rows
    against float64, including rows of the second range."""
    from dfx import ops
    M, N, K = (1 << 19) + 130, 256, 1024
    x = torch.randn(M, K, device="cuda")
    w = torch.randn(N, K, device="cuda") / 32
    b = torch.randn(N, device="cuda")
    assert x.numel() * 4 >= 1 << 31
    y = ops.linear(x, w, b, relu=True)
    rows = torch.tensor([0, 1, 4095, (1 << 19) - 129, (1 << 19) - 1, 1 << 19, (1 << 19) + 129, M - 1], device="cuda")
    want = (x[rows].double() @ w.double().t() + b.double()).relu()
    assert (y[rows].double() - want).abs().max().item() < 4e-6 * K ** 0.5


def test_conv_plan_runs_in_image_ranges_and_rejects_what_it_does_not_cover(monkeypatch):
    """The Winograd kernel indexes its input with 32-bit element offsets (2^30 elements per launch): ConvPlan splits
    larger batches into image ranges (limit lowered here: same bits as one launch).  Asymmetric strides / paddings,
    grouped convolutions and non-zero padding modes raise instead of computing something else."""
    from dfx import ops
    g = torch.Generator().manual_seed(5)
    x = torch.randn(5, 16, 20, 28, generator=g).cuda()
    wt = (torch.randn(64, 16, 3, 3, generator=g) / 12).cuda()
    plan = ops.ConvPlan(wt, torch.randn(64, generator=g).cuda(), 1, 1, 1, "relu")
    assert plan.algo == "wino"
    whole = plan(x)
    monkeypatch.setattr(ops.ConvPlan, "WINO_MAX_ELEMENTS", 2 * 16 * 20 * 28 + 1)     # -> 2 images per launch
    assert torch.equal(plan(x), whole)
    for bad in (dict(stride=(1, 2)), dict(padding=(1, 0)), dict(dilation=(2, 1)), dict(groups=2), dict(padding_mode="reflect")):
        with pytest.raises(RuntimeError):
            ops.ConvPlan(wt, None, **bad)


@pytest.mark.parametrize("K1,K2,Co,H,W,N", [(64, 64, 256, 20, 34, 3), (512, 1024, 2048, 10, 14, 2), (16, 32, 64, 6, 10, 1),
                                            (128, 48, 200, 7, 12, 2)])
def test_conv1x1_pair_matches_fp64(K1, K2, Co, H, W, N):
    """relu(W x [x1 ; x2] + b) - a bottleneck's conv3 and its stride-1 projection shortcut as ONE product over the
    concatenated channels (dfx_conv1x1_pair_f32: two-segment [K,N] operand, the K loop switches descriptor at K1)."""
    from dfx import ops
    g = torch.Generator().manual_seed(K1 + K2 + Co)
    x1 = torch.randn(N, K1, H, W, generator=g).cuda()
    x2 = torch.randn(N, K2, H, W, generator=g).cuda()
    w = (torch.randn(Co, K1 + K2, generator=g) / (K1 + K2) ** 0.5).cuda()
    b = torch.randn(Co, generator=g).cuda()
    got = ops.conv1x1_pair(x1, x2, w, b, relu=True)
    want = (torch.einsum("ok,nkhw->nohw", w.double(), torch.cat([x1, x2], 1).double()) + b.double().view(1, -1, 1, 1)).relu()
    assert got.shape == (N, Co, H, W)
    assert (got.double() - want).abs().max().item() < 4e-6 * (K1 + K2) ** 0.5
    assert torch.equal(ops.conv1x1_pair(x1, x2, w, None, relu=False) + b.view(1, -1, 1, 1),
                       ops.conv1x1(torch.cat([x1, x2], 1), w, None) + b.view(1, -1, 1, 1))


@pytest.mark.parametrize("M,K,act,act_first,res,add,blocked", [
    (4200, 256, None, False, True, False, False), (4200, 1024, None, False, True, False, False),
    (1200, 256, "gelu", True, True, False, False), (333, 256, None, False, False, False, False),
    (8400, 256, None, False, True, False, True), (300, 2048, None, False, True, False, False),
    (130, 64, "relu", False, True, True, False)])
def test_linear_with_layernorm_epilogue_matches_fp64(M, K, act, act_first, res, add, blocked, monkeypatch):
    """LayerNorm(residual + act(x W^T + b)) in ONE launch (dfx_linear_ln_f32: 64 x 256 tile, row statistics by wave
    reductions in the epilogue) against the float64 formulation, incl. the activation-before-residual order of the fusion
    blocks, no residual, the K-block-major A operand the level kernel writes, the ``add`` prologue."""
    from dfx import ops
    monkeypatch.setattr(ops, "_FUSE_LN", True)              # (off by default: slower than two launches, see dfx/ops.py)
    monkeypatch.setattr(ops, "_FUSE_LN_MIN_ROWS", 0)
    g = torch.Generator().manual_seed(M + K)
    x = torch.randn(M, K, generator=g).cuda()
    w = (torch.randn(256, K, generator=g) / K ** 0.5).cuda()
    b = torch.randn(256, generator=g).cuda()
    r = torch.randn(M, 256, generator=g).cuda() if res else None
    a = torch.randn(M, K, generator=g).cuda() if add else None
    norm = torch.nn.LayerNorm(256).cuda()
    with torch.no_grad():
        norm.weight.copy_(torch.randn(256, generator=g) * 0.2 + 1)
        norm.bias.copy_(torch.randn(256, generator=g) * 0.2)
    xin = x.view(M, K // 4, 4).permute(1, 0, 2).contiguous() if blocked else x
    got = ops.linear(xin, w, b, act=act, residual=r, add=a, norm=norm, act_first=act_first, x_blocked=blocked)
    y = (x.double() + (0 if a is None else a.double())) @ w.double().t() + b.double()
    fn = {None: lambda t: t, "relu": torch.relu, "gelu": torch.nn.functional.gelu}[act]
    if act_first:
        y = fn(y)
    if r is not None:
        y = y + r.double()
    if not act_first:
        y = fn(y)
    want = torch.nn.functional.layer_norm(y, (256,), norm.weight.double(), norm.bias.double(), norm.eps)
    assert got.shape == (M, 256)
    assert (got.double() - want).abs().max().item() < 2e-5
    # and the same bits as the two-launch formulation up to rounding of the statistics
    sep = ops.add_layernorm(ops.linear(xin, w, b, act=act if act_first else None, add=a, x_blocked=blocked) if (act_first or act is None)
                            else ops.linear(xin, w, b, act=act, residual=r, add=a, x_blocked=blocked),
                            r if (act_first or act is None) else None, norm)
    assert (got - sep).abs().max().item() < 2e-5


@pytest.mark.parametrize("M,n,w,K,masked", [(16800, 6, 256, 256, True), (4200, 3, 256, 256, False), (1200, 3, 512, 256, False),
                                           (130, 2, 128, 64, True)])
def test_linear_stack_of_wide_column_blocks(M, n, w, K, masked):
    """col_block = w, a multiple of 128: n Linears over the same rows stacked along N, one launch, every result a contiguous
    [M, w] tensor (the decoder layers' value projections, models/ops/modules/ms_deform_attn.py:project_values) - against the
    n separate Linears and fp64."""
    from dfx import ops
    g = torch.Generator().manual_seed(M + n + w)
    x = torch.randn(M, K, generator=g).cuda()
    ws = [(torch.randn(w, K, generator=g) / K ** 0.5).cuda() for _ in range(n)]
    bs = [torch.randn(w, generator=g).cuda() for _ in range(n)]
    mask = (torch.rand(M, generator=g) > 0.7).cuda() if masked else None
    out = ops.linear(x, torch.cat(ws, 0).contiguous(), torch.cat(bs, 0).contiguous(), row_mask=mask, col_block=w)
    assert out.shape == (n, M, w)
    for i in range(n):
        alone = ops.linear(x, ws[i], bs[i], row_mask=mask)
        want = x.double() @ ws[i].double().t() + bs[i].double()
        if masked:
            want = want.masked_fill(mask[:, None], 0.0)
        assert (out[i].double() - want).abs().max().item() < 4e-6 * K ** 0.5
        assert (out[i] - alone).abs().max().item() < 4e-6 * K ** 0.5
