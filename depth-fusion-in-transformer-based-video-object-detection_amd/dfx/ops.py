"""torch-tensor front ends of the C ABI (device pointers + torch's current stream).

Mirrors the host half of the reference's CUDA op
(/root/reference/models/ops/src/cuda/ms_deform_attn_cuda.cu:20-153): the same
argument checks, the same dimension derivation (L from spatial_shapes.size(0), Lq from
sampling_loc.size(1), P from sampling_loc.size(4)), fresh output tensors.
"""
import os

import torch

from . import _lib

_SUFFIX = {torch.float32: "f32", torch.float64: "f64"}
ACT = {None: 0, "none": 0, "relu": 1, "gelu": 2}

# Single-level attention with many queries (encoder, depth fusion) runs on the level-in-LDS kernel
# (csrc/msda_level.hip) when the level fits the CU's LDS; DFX_MSDA_LEVEL=0 keeps the wave-per-query
# kernel (A/B measurements).  LEVEL_MIN_QUERIES: below it staging the level costs more than it saves.
USE_LEVEL_KERNEL = os.environ.get("DFX_MSDA_LEVEL", "1") == "1"
LEVEL_MIN_QUERIES = 1024
# convolutions of <= 4 input channels from an LDS-resident input tile (csrc/conv_tile.hip); 0: on the implicit GEMM (A/B runs)
USE_TILE_CONV = os.environ.get("DFX_TILE_CONV", "1") == "1"
# msda_fused_forward takes operands in the reference layouts, where the level kernel is slower than the
# wave-per-query kernel (33 vs 26 us at the encoder geometry): it uses it only when this is set (parity tests).
LEVEL_ON_REFERENCE_LAYOUTS = False

# Measurement hook (bench.py): profile_start() makes every fused MSDA kernel stamp its own begin / end
# timestamps (include/dfx_msda.h, dfx_profile_*); profile_stop() returns [(seconds, algorithmic_bytes,
# Lq, S), ...] for the launches in between.
def profile_start():
    _lib.load().dfx_profile_enable(1)


def reload_tuning():
    """The library reads its DFX_* environment switches once per process (csrc/dfx_common.h:Tuning); call this after
    changing one of them in a running process (tests, A/B tools)."""
    _lib.load().dfx_tuning_reload()


def profile_stop(cap=65536):
    import ctypes
    lib = _lib.load()
    lib.dfx_profile_enable(0)
    ms = (ctypes.c_float * cap)()
    nb = (ctypes.c_long * cap)()
    lq = (ctypes.c_int * cap)()
    s = (ctypes.c_int * cap)()
    n = lib.dfx_profile_drain(ctypes.cast(ms, ctypes.c_void_p), ctypes.cast(nb, ctypes.c_void_p),
                              ctypes.cast(lq, ctypes.c_void_p), ctypes.cast(s, ctypes.c_void_p), cap)
    return [(ms[i] * 1e-3, nb[i], lq[i], s[i]) for i in range(n)]


def _require(cond, msg):
    if not cond:
        raise RuntimeError(msg)


def _check_inputs(named):
    # (messages are only formatted on failure: this runs for every launch of the path)
    for name, t in named:
        if not t.is_contiguous():
            raise RuntimeError(f"{name} tensor has to be contiguous")
    for name, t in named:
        # the reference raises "Not implemented on the CPU" (ms_deform_attn.h:38,60)
        if not t.is_cuda:
            raise RuntimeError(f"Not implemented on the CPU ({name} must be a CUDA tensor)")
    dev = named[0][1].device
    for name, t in named:
        if t.device != dev:
            raise RuntimeError(f"{name} is on {t.device}, expected {dev}")


def _dims(value, spatial_shapes, sampling_loc, im2col_step):
    N, S, M, D = value.shape
    L = spatial_shapes.shape[0]
    Lq = sampling_loc.shape[1]
    P = sampling_loc.shape[4]
    step = min(N, im2col_step)
    _require(step > 0 and N % step == 0, f"batch({N}) must divide im2col_step({step})")
    return N, S, M, D, L, Lq, P


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream(dev):
    """Raw handle of the current stream of `dev` (host time matters here: ~450 launches per 13 ms step of a 4-frame block)."""
    if _raw_stream is not None:
        return _raw_stream(dev.index if dev.index is not None else torch.cuda.current_device())
    return torch.cuda.current_stream(dev).cuda_stream


class _NoSwitch:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_SWITCH = _NoSwitch()


def _on(dev):
    """Context that makes `dev` the current device - a no-op object when it already is (one process per GPU: always)."""
    if dev.index is None or dev.index == torch.cuda.current_device():
        return _NO_SWITCH
    return torch.cuda.device(dev)


def msda_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step=64):
    lib = _lib.load()
    _check_inputs([("value", value), ("spatial_shapes", spatial_shapes),
                   ("level_start_index", level_start_index), ("sampling_loc", sampling_loc),
                   ("attn_weight", attn_weight)])
    _require(value.dtype in _SUFFIX, f"ms_deform_attn_forward not implemented for {value.dtype}")
    _require(sampling_loc.dtype == value.dtype and attn_weight.dtype == value.dtype,
             "value, sampling_loc and attn_weight must share one dtype")
    _require(spatial_shapes.dtype == torch.int64 and level_start_index.dtype == torch.int64,
             "spatial_shapes and level_start_index must be int64")
    N, S, M, D, L, Lq, P = _dims(value, spatial_shapes, sampling_loc, im2col_step)
    _require(sampling_loc.numel() >= N * Lq * M * L * P * 2 and attn_weight.numel() >= N * Lq * M * L * P,
             "sampling_loc / attn_weight smaller than N*Lq*M*L*P")
    out = torch.empty((N, Lq, M * D), dtype=value.dtype, device=value.device)
    fn = getattr(lib, "dfx_msda_forward_" + _SUFFIX[value.dtype])
    with _on(value.device):
        rc = fn(value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(),
                sampling_loc.data_ptr(), attn_weight.data_ptr(), N, S, M, D, L, Lq, P,
                out.data_ptr(), _stream(value.device))
    _lib.check(rc, "ms_deform_attn_forward")
    return out


def msda_backward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output,
                  im2col_step=64):
    lib = _lib.load()
    _check_inputs([("value", value), ("spatial_shapes", spatial_shapes),
                   ("level_start_index", level_start_index), ("sampling_loc", sampling_loc),
                   ("attn_weight", attn_weight), ("grad_output", grad_output)])
    _require(value.dtype in _SUFFIX, f"ms_deform_attn_backward not implemented for {value.dtype}")
    _require(sampling_loc.dtype == value.dtype and attn_weight.dtype == value.dtype
             and grad_output.dtype == value.dtype, "all floating inputs must share one dtype")
    N, S, M, D, L, Lq, P = _dims(value, spatial_shapes, sampling_loc, im2col_step)
    _require(grad_output.numel() == N * Lq * M * D, "grad_output has the wrong size")
    grad_value = torch.zeros_like(value)
    grad_loc = torch.zeros_like(sampling_loc)
    grad_aw = torch.zeros_like(attn_weight)
    fn = getattr(lib, "dfx_msda_backward_" + _SUFFIX[value.dtype])
    with _on(value.device):
        rc = fn(value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(),
                sampling_loc.data_ptr(), attn_weight.data_ptr(), grad_output.data_ptr(),
                N, S, M, D, L, Lq, P, grad_value.data_ptr(), grad_loc.data_ptr(), grad_aw.data_ptr(),
                _stream(value.device))
    _lib.check(rc, "ms_deform_attn_backward")
    return [grad_value, grad_loc, grad_aw]


def fused_supported(value, M, D, L, P, Lr):
    """Geometry the fused front-end kernel covers (else callers take the unfused op)."""
    return (value.is_cuda and value.dtype == torch.float32 and M == 8 and D == 32 and P == 4
            and 1 <= L <= 4 and (Lr == L or L == 1))


def msda_fused_forward(value, spatial_shapes, level_start_index, reference_points, qproj, n_levels, n_points):
    """softmax + location arithmetic + sampling in one launch (include/dfx_msda.h,
    dfx_msda_fused_forward_f32).

    value            [N,S,M,D] fp32, contiguous
    reference_points [N,Lq,Lr,2|4]
    qproj            [N,Lq,3*M*L*P] = one row per query holding the raw sampling_offsets
                     Linear output (M*L*P*2 floats) followed by the raw attention_weights
                     Linear output (M*L*P floats)
    -> [N,Lq,M*D]
    """
    lib = _lib.load()
    reference_points = reference_points.contiguous()
    _check_inputs([("value", value), ("spatial_shapes", spatial_shapes),
                   ("level_start_index", level_start_index), ("reference_points", reference_points),
                   ("qproj", qproj)])
    N, S, M, D = value.shape
    L, P = n_levels, n_points
    _require(spatial_shapes.shape[0] == L, "spatial_shapes rows must equal n_levels")
    Lq, Lr, ref_dim = reference_points.shape[1], reference_points.shape[2], reference_points.shape[3]
    mlp = M * L * P
    _require(qproj.shape == (N, Lq, 3 * mlp) and qproj.dtype == torch.float32, "qproj has the wrong shape/dtype")
    _require(reference_points.dtype == torch.float32 and reference_points.shape[0] == N, "bad reference_points")
    out = torch.empty((N, Lq, M * D), dtype=value.dtype, device=value.device)
    base = qproj.data_ptr()
    host = getattr(spatial_shapes, "_dfx_host", None)
    level = (USE_LEVEL_KERNEL and LEVEL_ON_REFERENCE_LAYOUTS and host is not None and L == 1 and Lr == 1 and M == 8 and D == 32
             and P == 4 and S == host[0][0] * host[0][1] and Lq >= LEVEL_MIN_QUERIES
             and lib.dfx_msda_fused_level_fits(host[0][0], host[0][1]))
    with _on(value.device):
        if level:  # the whole level lives in LDS
            import ctypes
            ly = _lib.LevelLayout(S * 256, 256, 32, 8, 4, 3 * mlp, 8, 3 * mlp, 4, 256, 32, 8, 4)   # reference layouts
            rc = lib.dfx_msda_fused_level_forward_f32(
                value.data_ptr(), reference_points.data_ptr(), ref_dim, base, base + 2 * mlp * 4, ctypes.byref(ly),
                N, host[0][0], host[0][1], Lq, out.data_ptr(), _stream(value.device))
        else:
            rc = lib.dfx_msda_fused_forward_f32(
                value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(),
                reference_points.data_ptr(), ref_dim, Lr,
                base, 3 * mlp, base + 2 * mlp * 4, 3 * mlp,
                N, S, M, D, L, Lq, P, out.data_ptr(), _stream(value.device))
    _lib.check(rc, "msda_fused_forward")
    return out


def level_supported(value_like, H, W, Lq, n_heads, head_dim, n_levels, n_points, n_ref_levels):
    """Single-level attention the level-in-LDS kernel (csrc/msda_level.hip) should run: its
    geometry, a level that fits the CU's LDS, and enough queries to pay for staging the level."""
    return (USE_LEVEL_KERNEL and value_like.is_cuda and value_like.dtype == torch.float32 and n_heads == 8
            and head_dim == 32 and n_points == 4 and n_levels == 1 and n_ref_levels == 1
            and Lq >= LEVEL_MIN_QUERIES and bool(_lib.load().dfx_msda_fused_level_fits(int(H), int(W))))


def msda_level_forward(value_blk, reference_points, qproj_blk, N, H, W):
    """Fused single-level MSDA on operands in the block-major layouts linear(col_block=...) writes and
    linear(x_blocked=True) reads (include/dfx_msda.h, dfx_msda_fused_level_forward_f32):

    value_blk        [64, N*H*W, 4]  4-channel chunk k of every token (value_proj, col_block=4)
    reference_points [N, Lq, 1, 2|4]
    qproj_blk        [8, N*Lq, 12]   per head: 4 points x (x, y) offsets, then the 4 logits (col_block=12 of the
                                     head-interleaved sampling_offsets / attention_weights Linear)
    -> [64, N*Lq, 4]  the sampled values, 4-channel chunk k of every query (output_proj's x_blocked operand)
    """
    import ctypes
    lib = _lib.load()
    reference_points = reference_points.contiguous()
    _check_inputs([("value_blk", value_blk), ("reference_points", reference_points), ("qproj_blk", qproj_blk)])
    S = H * W
    Lq, ref_dim = reference_points.shape[1], reference_points.shape[3]
    _require(value_blk.shape == (64, N * S, 4) and value_blk.dtype == torch.float32, "value_blk must be [64, N*H*W, 4] fp32")
    _require(qproj_blk.shape == (8, N * Lq, 12) and qproj_blk.dtype == torch.float32, "qproj_blk must be [8, N*Lq, 12] fp32")
    _require(reference_points.shape[0] == N and reference_points.shape[2] == 1 and reference_points.dtype == torch.float32,
             "reference_points must be [N, Lq, 1, 2|4] fp32")
    out = torch.empty((64, N * Lq, 4), dtype=torch.float32, device=value_blk.device)
    base = qproj_blk.data_ptr()
    ns, nq = N * S, N * Lq
    ly = _lib.LevelLayout(S * 4, 4, 32 * ns, 8 * ns, 4 * ns, 12, 12 * nq, 12, 12 * nq, 4, 32 * nq, 8 * nq, 4 * nq)
    with _on(value_blk.device):
        rc = lib.dfx_msda_fused_level_forward_f32(
            value_blk.data_ptr(), reference_points.data_ptr(), ref_dim, base, base + 32, ctypes.byref(ly),
            N, H, W, Lq, out.data_ptr(), _stream(value_blk.device))
    _lib.check(rc, "msda_level_forward")
    return out


def roi_align(inp, rois, output_size, spatial_scale, sampling_ratio, aligned=True, channels_last=False):
    """RoIAlign (avg, fixed sampling grid) through include/dfx_roi.h.

    channels_last=False: inp [N,C,H,W] contiguous -> [K,C,ph,pw]
    channels_last=True : inp [N,H,W,C] contiguous -> [K,ph*pw,C]   (token-major memory, no transpose)
    rois [K,5] = (batch index, x1, y1, x2, y2)
    """
    lib = _lib.load()
    ph, pw = (output_size, output_size) if isinstance(output_size, int) else output_size
    rois = rois.contiguous().float()
    _check_inputs([("input", inp), ("rois", rois)])
    _require(inp.dtype == torch.float32, "roi_align is implemented for float32")
    _require(rois.dim() == 2 and rois.shape[1] == 5, "rois must be [K,5]")
    K = rois.shape[0]
    if channels_last:
        N, H, W, C = inp.shape
        out = torch.empty((K, ph * pw, C), dtype=inp.dtype, device=inp.device)
        fn = lib.dfx_roi_align_nhwc_f32
    else:
        N, C, H, W = inp.shape
        out = torch.empty((K, C, ph, pw), dtype=inp.dtype, device=inp.device)
        fn = lib.dfx_roi_align_nchw_f32
    with _on(inp.device):
        rc = fn(inp.data_ptr(), rois.data_ptr(), N, C, H, W, K, ph, pw, float(spatial_scale),
                int(sampling_ratio), int(bool(aligned)), out.data_ptr(), _stream(inp.device))
    _lib.check(rc, "roi_align")
    return out


def bias_act_(x, bias, residual=None, relu=True):
    """In place on ``x`` [N,C,*spatial] (contiguous NCHW): x = relu?(x + bias[c] (+ residual)).
    One HBM pass for what the reference runs as FrozenBatchNorm2d (after folding its scale into
    the convolution weights) + residual add + ReLU (include/dfx_fused.h)."""
    lib = _lib.load()
    named = [("x", x), ("bias", bias)] + ([("residual", residual)] if residual is not None else [])
    _check_inputs(named)
    _require(x.dtype == torch.float32 and bias.dtype == torch.float32, "bias_act_ is implemented for float32")
    N, C = x.shape[0], x.shape[1]
    _require(bias.numel() == C, "bias must have one entry per channel")
    if residual is not None:
        _require(residual.shape == x.shape and residual.dtype == x.dtype, "residual must match x")
    hw = x.numel() // max(N * C, 1)
    with _on(x.device):
        rc = lib.dfx_bias_act_nchw_f32(x.data_ptr(), bias.data_ptr(), 0 if residual is None else residual.data_ptr(),
                                       x.data_ptr(), N, C, hw, int(bool(relu)), _stream(x.device))
    _lib.check(rc, "bias_act_")
    return x


def _ptr(t):
    return 0 if t is None else t.data_ptr()


_GEMM_MAX_BYTES = 1 << 31          # tests lower it to exercise the row-range path on small tensors
# LayerNorm in the GEMM epilogue (dfx_linear_ln_f32) - OFF: measured slower than GEMM + add_layernorm on this chip (round 3,
# profiles/r03_linear_ln.txt: the 64 x 256 tile that owns whole rows runs its K loop with 36 % more LDS-DMA pieces per MFMA than
# the 128 x 128 tile and leaves 19 workgroups for the 1200-row Linears of a 4-frame block: spatial stage 16.80 -> 16.97 ms with
# it on the 134 400-row Linears only, 16.7 -> 17.3 ms on all of them, the 4-frame rank step 16.65 -> 18.9 ms).  DFX_LINEAR_LN=1
# turns it on for A/B runs; the callers' ``norm=`` plumbing then costs nothing when it is off.
# (A/B of round 3: the residual in the GEMM epilogue instead of in the LayerNorm pass moves 0.06 ms of 15.5 in the spatial stage -
# inside the noise; off)
_RESIDUAL_IN_GEMM = os.environ.get("DFX_LN_RESIDUAL_IN_GEMM", "0") == "1"
_FUSE_LN = os.environ.get("DFX_LINEAR_LN", "0") == "1"
_FUSE_LN_MIN_ROWS = int(os.environ.get("DFX_LINEAR_LN_MIN_ROWS", "32768"))


_SPLITK_MIN_K = int(os.environ.get("DFX_SPLITK_MIN_K", "2048"))      # (A/B aids)
_SPLITK_RANGE = int(os.environ.get("DFX_SPLITK_RANGE", "512"))


def _split_k(M, N, K):
    """Number of K ranges for a GEMM with few output tiles and a long K (0 = no split): as many as keep the
    128 x 128 (or 64 x 128 for small M) tiles x ranges within one resident round of workgroups (3 per CU for the
    large tile, 4 for the small one), at least 32 K-steps per range."""
    if K < _SPLITK_MIN_K:
        return 0
    if M >= 128 and N >= 128:
        tiles, slots = -(-M // 128) * -(-N // 128), 768
    else:
        tiles, slots = -(-M // 64) * -(-N // 128), 1024
    if tiles * 2 > slots:
        return 0
    return max(1, min(K // _SPLITK_RANGE, slots // tiles))


def linear(x, weight, bias=None, relu=False, residual=None, add=None, row_mask=None, col_block=0, x_blocked=False,
           act=None, norm=None, act_first=False):
    """y = act((x (+ add)) @ weight.T + bias (+ residual)), rows where row_mask is True set to 0.
    act: None / "relu" / "gelu" (``relu=True`` is the older spelling of act="relu").
    The hand-written fp32 MFMA GEMM (include/dfx_gemm.h) standing in for nn.Linear with its
    neighbours fused: the ``src + pos`` query add, the bias, ReLU, the residual add and
    value_proj's masked_fill.  x [..., K] contiguous, weight [N, K] -> [..., N].

    norm: an nn.LayerNorm(256) applied to the result rows in the same launch (include/dfx_gemm.h, dfx_linear_ln_f32):
    y = norm(residual + act(...)) with ``act_first`` (the activation before the residual add), else norm(act(... + residual));
    needs N == 256, no row_mask, no col_block.

    col_block = w > 0 stores the result column-block-major instead: [N / w, rows, w] (the layout
    msda_level_forward reads), N a multiple of w.  x_blocked: x is K-block-major [K/4, rows, 4] (the
    layout msda_level_forward writes); the result is then [rows, N]."""
    lib = _lib.load()
    N = weight.shape[0]
    if x_blocked:
        _require(x.dim() == 3 and x.shape[2] == 4 and add is None, "x_blocked: x must be [K/4, rows, 4], no add")
        K, M, x2 = x.shape[0] * 4, x.shape[1], x
    else:
        K = x.shape[-1]
        x2 = x.reshape(-1, K)
        M = x2.shape[0]
    named = [("x", x2), ("weight", weight)]
    for nm, t in (("bias", bias), ("residual", residual), ("add", add), ("row_mask", row_mask)):
        if t is not None:
            named.append((nm, t))
    _check_inputs(named)
    _require(x2.dtype == torch.float32 and weight.dtype == torch.float32, "linear is implemented for float32")
    _require(weight.shape[1] == K and K % 4 == 0, "weight must be [N, K] with K a multiple of 4")
    if add is not None:
        _require(add.shape == x.shape, "add must match x")
    if col_block:
        _require(N % col_block == 0 and residual is None, "col_block must divide N; no residual in this layout")
        out = torch.empty((N // col_block, M, col_block), dtype=x.dtype, device=x.device)
    elif x_blocked:
        out = torch.empty((M, N), dtype=x.dtype, device=x.device)
    else:
        out = torch.empty(x.shape[:-1] + (N,), dtype=x.dtype, device=x.device)
    if residual is not None:
        _require(residual.shape == out.shape or (x_blocked and residual.numel() == out.numel()),
                 "residual must match the output")
    if row_mask is not None:
        _require(row_mask.numel() == M, "row_mask must have one entry per row")
        row_mask = row_mask.reshape(-1).to(torch.uint8) if row_mask.dtype != torch.uint8 else row_mask.reshape(-1)
    code = ACT[act] if act is not None else int(bool(relu))
    if norm is not None:      # one contract for both routes below (separate LayerNorm pass / LayerNorm in the GEMM epilogue)
        _require(N == 256 and row_mask is None and not col_block and norm.weight.numel() == 256 and norm.weight.is_cuda,
                 "linear(norm=...): a LayerNorm over exactly 256 output columns, no row_mask / col_block")
    if norm is not None and (not _FUSE_LN or M < _FUSE_LN_MIN_ROWS):       # the LayerNorm as its own pass
        if (code and not act_first) or (not code and _RESIDUAL_IN_GEMM):
            # the residual rides in the GEMM's epilogue (prefetched while the tile crosses LDS): the LayerNorm pass then reads one
            # tensor instead of two.  Same sum, same order: (x W^T + b) + residual
            y = linear(x, weight, bias, residual=residual, add=add, x_blocked=x_blocked, act=act, relu=relu)
            return add_layernorm(y, None, norm)
        y = linear(x, weight, bias, add=add, x_blocked=x_blocked, act=act, relu=relu)
        return add_layernorm(y, None if residual is None else residual.reshape(y.shape), norm)
    if norm is not None:
        _require(M * K * 4 < _GEMM_MAX_BYTES, "linear(norm=...): operand of 2 GiB or more")
        with _on(x.device):
            rc = lib.dfx_linear_ln_f32(x2.data_ptr(), _ptr(add), K, M * 4 if x_blocked else 0, weight.data_ptr(), K, _ptr(bias),
                                       _ptr(residual), N, norm.weight.data_ptr(), norm.bias.data_ptr(), float(norm.eps),
                                       out.data_ptr(), N, M, K, code, int(bool(act_first)), _stream(x.device))
        _lib.check(rc, "linear + LayerNorm")
        return out
    splits = _split_k(M, N, K) if (add is None and row_mask is None and not col_block and not x_blocked and N % 4 == 0) else 0
    with _on(x.device):
        if splits > 1:
            ws = torch.empty((splits, M, N), dtype=torch.float32, device=x.device)
            rc = lib.dfx_gemm_splitk_f32(x2.data_ptr(), K, weight.data_ptr(), K, 0, _ptr(bias), 0, _ptr(residual), N,
                                         out.data_ptr(), N, M, N, K, code, splits, ws.data_ptr(), _stream(x.device))
            _lib.check(rc, "linear (split-K)")
            return out
        # The kernel addresses an operand with 32-bit byte offsets (buffer loads): A, and the residual, must stay below
        # 2 GiB per call.  More rows than that (long clips of multi-scale token maps) go through in row ranges; every
        # layout here is row-separable (strides are explicit arguments), so a range is the same call on offset pointers.
        rows_max = (_GEMM_MAX_BYTES - 1) // (4 * max(K, N if residual is not None else 1))
        if x_blocked:
            _require(M * K * 4 < _GEMM_MAX_BYTES, "linear: a K-block-major x of 2 GiB or more is not supported")
            rows_max = M
        rows_max = max(128, rows_max // 128 * 128)
        out_row = int(col_block) if col_block else N            # elements a row advances the output pointer by
        for r0 in range(0, M, rows_max):
            r1 = min(M, r0 + rows_max)
            off = lambda t, per_row: 0 if t is None else t.data_ptr() + r0 * per_row * t.element_size()  # noqa: E731
            rc = lib.dfx_gemm_f32(off(x2, 4 if x_blocked else K), off(add, K), K, 0, weight.data_ptr(), K, 0, 0, _ptr(bias), 0,
                                  off(residual, N), N, 0, off(row_mask, 1), 0, off(out, out_row), N, 0, r1 - r0, N, K, 1,
                                  code, int(col_block), M * int(col_block), M * 4 if x_blocked else 0,
                                  _stream(x.device))
            _lib.check(rc, "linear")
    return out


def conv1x1(x, weight, bias=None, residual=None, relu=False, stride=1):
    """1x1 convolution on NCHW as a batched GEMM W[Co,Ci] x X_n[Ci,HW] with the folded-BN bias,
    the residual add and ReLU fused into the epilogue.  x [N,Ci,H,W], weight [Co,Ci(,1,1)]."""
    lib = _lib.load()
    if stride != 1:
        x = x[:, :, ::stride, ::stride].contiguous()
    Nb, Ci, H, W = x.shape
    Co = weight.shape[0]
    w2 = weight.reshape(Co, -1)
    named = [("x", x), ("weight", w2)] + [(n, t) for n, t in (("bias", bias), ("residual", residual)) if t is not None]
    _check_inputs(named)
    _require(x.dtype == torch.float32 and w2.shape[1] == Ci, "conv1x1: fp32, weight [Co,Ci]")
    HW = H * W
    _require(Ci % 4 == 0 and HW % 4 == 0, "conv1x1 needs Ci and H*W to be multiples of 4")
    out = torch.empty((Nb, Co, H, W), dtype=x.dtype, device=x.device)
    if residual is not None:
        _require(residual.shape == out.shape, "residual must match the output")
    with _on(x.device):
        rc = lib.dfx_gemm_f32(w2.data_ptr(), 0, Ci, 0, x.data_ptr(), HW, Ci * HW, 1, _ptr(bias), 1,
                              _ptr(residual), HW, Co * HW, 0, 0, out.data_ptr(), HW, Co * HW, Co, HW, Ci, Nb,
                              int(bool(relu)), 0, 0, 0, _stream(x.device))
    _lib.check(rc, "conv1x1")
    return out


def conv1x1_pair(x1, x2, weight, bias=None, relu=False):
    """relu?(weight x [x1 ; x2] + bias): two NCHW inputs of one map size concatenated along the channels inside the
    product (include/dfx_gemm.h, dfx_conv1x1_pair_f32) - a bottleneck's last 1x1 convolution and its stride-1
    projection shortcut in one GEMM.  x1 [N,K1,H,W], x2 [N,K2,H,W], weight [Co,K1+K2] -> [N,Co,H,W]."""
    lib = _lib.load()
    _check_inputs([("x1", x1), ("x2", x2), ("weight", weight)] + ([("bias", bias)] if bias is not None else []))
    Nb, K1, H, W = x1.shape
    K2 = x2.shape[1]
    Co = weight.shape[0]
    _require(x1.dtype == torch.float32 and x2.shape == (Nb, K2, H, W) and weight.shape == (Co, K1 + K2),
             "conv1x1_pair: x1 [N,K1,H,W], x2 [N,K2,H,W], weight [Co,K1+K2], fp32")
    out = torch.empty((Nb, Co, H, W), dtype=x1.dtype, device=x1.device)
    with _on(x1.device):
        rc = lib.dfx_conv1x1_pair_f32(weight.data_ptr(), x1.data_ptr(), K1 * H * W, K1, x2.data_ptr(), K2 * H * W, K2,
                                      _ptr(bias), out.data_ptr(), Co * H * W, Co, H * W, Nb, int(bool(relu)),
                                      _stream(x1.device))
    _lib.check(rc, "conv1x1_pair")
    return out


def dynamic_conv(feats, params, norm1, norm2):
    """relu(norm2(relu(norm1(feats @ k1)) @ k2)) per RoI with k1, k2 cut from ``params`` (include/dfx_roi.h,
    dfx_dynamic_conv_f32): feats [K,R,256] contiguous, params [K, 2*256*64] (rows may be strided), norm1 / norm2
    nn.LayerNorm(64) / nn.LayerNorm(256).  -> [K,R,256]"""
    lib = _lib.load()
    _check_inputs([("feats", feats), ("norm1.weight", norm1.weight), ("norm2.weight", norm2.weight)])
    K, R, C = feats.shape
    dd = norm1.normalized_shape[0]
    _require(params.is_cuda and params.dim() == 2 and params.shape[0] == K and params.stride(1) == 1
             and params.shape[1] >= 2 * C * dd and params.dtype == torch.float32 and feats.dtype == torch.float32,
             "dynamic_conv: params must be [K, 2*C*dd] fp32 with contiguous rows")
    _require(norm2.normalized_shape[0] == C and norm1.eps == norm2.eps, "dynamic_conv: norm shapes / eps")
    out = torch.empty_like(feats)
    with _on(feats.device):
        rc = lib.dfx_dynamic_conv_f32(feats.data_ptr(), params.data_ptr(), params.stride(0), norm1.weight.data_ptr(),
                                      norm1.bias.data_ptr(), norm2.weight.data_ptr(), norm2.bias.data_ptr(),
                                      out.data_ptr(), K, R, C, dd, float(norm1.eps), _stream(feats.device))
    _lib.check(rc, "dynamic_conv")
    return out


def mha(q, k, v, heads, scale):
    """softmax(scale * q k^T) v per head (include/dfx_mha.h): q [B,Lq,E], k / v [B,Lk,E], E = 32*heads, fp32.
    The tensors may be column slices of a joint projection (last dimension contiguous).  -> [B,Lq,E]"""
    lib = _lib.load()
    for nm, t in (("q", q), ("k", k), ("v", v)):
        if not t.is_cuda:
            raise RuntimeError(f"{nm} must be a CUDA tensor (the fused attention has no CPU path)")
        _require(t.dim() == 3 and t.dtype == torch.float32 and t.stride(2) == 1 and t.shape[2] == 32 * heads,
                 f"mha: {nm} must be [B,L,32*heads] fp32 with a contiguous last dimension")
    B, Lq, E = q.shape
    Lk = k.shape[1]
    _require(k.shape == (B, Lk, E) and v.shape == (B, Lk, E), "mha: k and v must be [B,Lk,E]")
    out = torch.empty((B, Lq, E), dtype=torch.float32, device=q.device)
    with _on(q.device):
        rc = lib.dfx_mha_f32(q.data_ptr(), q.stride(0), q.stride(1), k.data_ptr(), k.stride(0), k.stride(1),
                             v.data_ptr(), v.stride(0), v.stride(1), out.data_ptr(), Lq * E, E, B, heads, Lq, Lk,
                             float(scale), _stream(q.device))
    _lib.check(rc, "mha")
    return out


def box_refine(delta, reference, eps=1e-5):
    """sigmoid(delta + inverse_sigmoid(reference)) on the first reference.shape[-1] (2 or 4) of the 4 box
    columns, sigmoid(delta) on the others, in one launch (include/dfx_fused.h).  delta [...,4]."""
    lib = _lib.load()
    delta, reference = delta.contiguous(), reference.contiguous()
    _check_inputs([("delta", delta), ("reference", reference)])
    rd = reference.shape[-1]
    _require(delta.shape[-1] == 4 and rd in (2, 4) and delta.shape[:-1] == reference.shape[:-1]
             and delta.dtype == torch.float32 and reference.dtype == torch.float32,
             "box_refine: delta [...,4], reference [...,2|4], fp32")
    out = torch.empty_like(delta)
    with _on(delta.device):
        rc = lib.dfx_box_refine_f32(delta.data_ptr(), reference.data_ptr(), rd, out.data_ptr(), delta.numel() // 4,
                                    float(eps), _stream(delta.device))
    _lib.check(rc, "box_refine")
    return out


def add_layernorm(x, residual, norm):
    """``norm(x + residual)`` for an nn.LayerNorm over the last dimension, in one pass
    (include/dfx_fused.h); residual may be None."""
    lib = _lib.load()
    C = x.shape[-1]
    x2 = x.contiguous()
    named = [("x", x2), ("weight", norm.weight), ("bias", norm.bias)]
    if residual is not None:
        residual = residual.contiguous()
        named.append(("residual", residual))
        _require(residual.shape == x2.shape, "residual must match x")
    _check_inputs(named)
    _require(x2.dtype == torch.float32 and norm.weight.numel() == C, "add_layernorm: fp32, LayerNorm over the last dim")
    out = torch.empty_like(x2)
    with _on(x2.device):
        rc = lib.dfx_add_layernorm_f32(x2.data_ptr(), _ptr(residual), norm.weight.data_ptr(), norm.bias.data_ptr(),
                                       out.data_ptr(), x2.numel() // C, C, float(norm.eps), _stream(x2.device))
    _lib.check(rc, "add_layernorm")
    return out


def bias_relu_maxpool(x, bias):
    """maxpool3x3/s2/p1(relu(x + bias[c])) on NCHW in one pass: the ResNet stem's epilogue
    (include/dfx_fused.h)."""
    lib = _lib.load()
    _check_inputs([("x", x), ("bias", bias)])
    _require(x.dtype == torch.float32 and x.dim() == 4 and bias.numel() == x.shape[1], "bias_relu_maxpool: fp32 NCHW")
    N, C, H, W = x.shape
    out = torch.empty((N, C, (H + 1) // 2, (W + 1) // 2), dtype=x.dtype, device=x.device)
    with _on(x.device):
        rc = lib.dfx_bias_relu_maxpool_f32(x.data_ptr(), bias.data_ptr(), out.data_ptr(), N, C, H, W, _stream(x.device))
    _lib.check(rc, "bias_relu_maxpool")
    return out



class ConvPlan:
    """One convolution prepared for the hand-written kernels (include/dfx_conv.h): 3x3 / stride 1 / "same"
    convolutions with Ci % 8 == 0 and Co % 64 == 0 run as fused Winograd F(2x2, 3x3) (weights pre-transformed
    on the GPU by dfx_wino_weights_f32), convolutions of at most 4 input channels (the 7x7/2 ResNet stem, the first
    DFormer convolution) as a direct convolution from an LDS-resident input tile (algo "tile", csrc/conv_tile.hip),
    everything else as an implicit GEMM over a tap table (weights re-ordered [Co, (ky, kx, ci)], K padded to a
    multiple of 16; the tile kernel takes the same weights).  ``scale`` (per output channel, e.g. the
    folded FrozenBatchNorm2d factor) is multiplied into the weights; ``bias`` and ``act`` run in the epilogue.
    ``weight`` is [Co, Ci, kh, kw] of an ungrouped convolution with zero padding: callers hand over ``conv.groups`` /
    ``conv.padding_mode`` (or check them) - anything else raises."""
    WINO_MAX_ELEMENTS = 1 << 30      # input elements per Winograd launch (32-bit offsets in the kernel); more go in image ranges

    def __init__(self, weight, bias=None, stride=1, padding=0, dilation=1, act=None, scale=None, algo=None, groups=1,
                 padding_mode="zeros"):
        _require(weight.is_cuda and weight.dtype == torch.float32 and weight.dim() == 4, "ConvPlan: fp32 CUDA weight [Co,Ci,kh,kw]")
        for nm, v in (("stride", stride), ("padding", padding), ("dilation", dilation)):
            _require(not isinstance(v, (tuple, list)) or len(set(int(e) for e in v)) == 1,
                     f"ConvPlan: {nm} must be the same along both axes (got {tuple(v) if isinstance(v, (tuple, list)) else v})")
        stride, padding, dilation = (v[0] if isinstance(v, (tuple, list)) else v for v in (stride, padding, dilation))
        Co, Ci, kh, kw = weight.shape
        _require(groups == 1 and padding_mode == "zeros", "ConvPlan: grouped convolutions / non-zero padding modes are not covered")
        _require(kh <= 32 and kw <= 32, "ConvPlan: kernel sides up to 32")
        self.Co, self.Ci, self.kh, self.kw = Co, Ci, kh, kw
        self.stride, self.padding, self.dilation, self.act = int(stride), int(padding), int(dilation), ACT[act]
        self.bias = None if bias is None else bias.detach().float().contiguous()
        wino_ok = kh == 3 and kw == 3 and stride == 1 and padding == dilation and Ci % 8 == 0 and Co % 64 == 0
        lib = _lib.load()
        tile_ok = bool(lib.dfx_conv2d_tile_fits(Ci, Co, kh, kw, int(stride), int(dilation)))
        self.algo = algo or ("wino" if wino_ok else "tile" if (tile_ok and Ci <= 4 and USE_TILE_CONV) else "igemm")
        _require(self.algo != "wino" or wino_ok, "ConvPlan: geometry not covered by the Winograd kernel")
        _require(self.algo != "tile" or tile_ok, "ConvPlan: geometry not covered by the tile kernel")
        w = weight.detach().contiguous()
        if self.algo == "wino":
            self.u = torch.empty(16 * Co * Ci, dtype=torch.float32, device=w.device)
            sc = None if scale is None else scale.detach().float().contiguous()
            with _on(w.device):
                rc = lib.dfx_wino_weights_f32(w.data_ptr(), _ptr(sc), self.u.data_ptr(), Co, Ci, _stream(w.device))
            _lib.check(rc, "wino_weights")
        else:
            if scale is not None:
                w = w * scale.detach().reshape(-1, 1, 1, 1)
            K = kh * kw * Ci
            Kpad = (K + 15) // 16 * 16
            _require(kh * kw <= 64, "ConvPlan: at most 64 taps")
            wp = torch.zeros(Co, Kpad, dtype=torch.float32, device=w.device)
            wp[:, :K] = w.permute(0, 2, 3, 1).reshape(Co, K)
            self.wp, self.Kpad, self.K, self._tabs = wp, Kpad, K, {}

    def _ktab(self, H, W, device):
        """Tap table of the implicit GEMM for an H x W input (include/dfx_conv.h): per k {tap index, byte offset}."""
        key = (H, W, str(device))
        if key not in self._tabs:
            ky, kx, ci = torch.meshgrid(torch.arange(self.kh), torch.arange(self.kw), torch.arange(self.Ci), indexing="ij")
            tab = torch.zeros((self.Kpad, 2), dtype=torch.int32)
            tab[:, 0] = -1
            tab[:self.K, 0] = (ky * self.kw + kx).reshape(-1).to(torch.int32)
            tab[:self.K, 1] = ((ci * (H * W) + ky * self.dilation * W + kx * self.dilation) * 4).reshape(-1).to(torch.int32)
            self._tabs[key] = tab.to(device)
        return self._tabs[key]

    def out_size(self, H, W):
        eff_h, eff_w = (self.kh - 1) * self.dilation + 1, (self.kw - 1) * self.dilation + 1
        return (H + 2 * self.padding - eff_h) // self.stride + 1, (W + 2 * self.padding - eff_w) // self.stride + 1

    def __call__(self, x):
        lib = _lib.load()
        _require(x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[1] == self.Ci,
                 "conv: x must be a CUDA fp32 tensor [N,Ci,H,W] (no CPU path)")
        N, _, H, W = x.shape
        # a channel slice of a wider NCHW tensor (x[:, :3] of an RGB-D clip) is read in place by the implicit GEMM
        sliced = (self.algo in ("igemm", "tile") and not x.is_contiguous() and x.stride(3) == 1 and x.stride(2) == W
                  and x.stride(1) == H * W and x.stride(0) >= self.Ci * H * W)
        image_stride = x.stride(0) if sliced else 0
        if not sliced:
            x = x.contiguous()
        Ho, Wo = self.out_size(H, W)
        y = torch.empty((N, self.Co, Ho, Wo), dtype=torch.float32, device=x.device)
        step = N
        if self.algo == "wino" and N * self.Ci * H * W >= self.WINO_MAX_ELEMENTS:
            step = max(1, (self.WINO_MAX_ELEMENTS - 1) // (self.Ci * H * W))
        if self.algo == "tile":           # 32-bit byte offsets over the images of a launch: below 4 GiB per launch
            step = max(1, min(N, ((1 << 30) - 4) // max(image_stride, self.Ci * H * W)))
        with _on(x.device):
            for n0 in range(0, N, step):
                n1 = min(N, n0 + step)
                xs, ys = x[n0:n1], y[n0:n1]
                if self.algo == "wino":
                    rc = lib.dfx_conv3x3_wino_f32(xs.data_ptr(), self.u.data_ptr(), _ptr(self.bias), ys.data_ptr(), n1 - n0, self.Ci,
                                                  H, W, self.Co, self.dilation, self.act, _stream(x.device))
                elif self.algo == "tile":
                    rc = lib.dfx_conv2d_tile_f32(xs.data_ptr(), self.wp.data_ptr(), _ptr(self.bias), ys.data_ptr(), n1 - n0, self.Ci,
                                                 H, W, self.Co, Ho, Wo, self.Kpad, self.kh, self.kw, self.stride, self.padding,
                                                 self.act, image_stride, _stream(x.device))
                else:
                    rc = lib.dfx_conv2d_igemm_f32(xs.data_ptr(), self.wp.data_ptr(), self._ktab(H, W, x.device).data_ptr(),
                                                  _ptr(self.bias), ys.data_ptr(), n1 - n0, self.Ci, H, W, self.Co, Ho, Wo, self.Kpad,
                                                  self.kh, self.kw, self.stride, self.padding, self.dilation, self.act,
                                                  image_stride, _stream(x.device))
                _lib.check(rc, "conv " + self.algo)
        return y


def group_norm(x, norm, tokens_out=False):
    """nn.GroupNorm ``norm`` applied to x [N,C,H,W] (contiguous NCHW) in two streaming launches
    (include/dfx_fused.h, dfx_group_norm_f32).  tokens_out: the result is written token-major [N,H*W,C] and
    returned as an NCHW-shaped VIEW of that memory, so ``y.flatten(2).transpose(1, 2)`` - what the transformer
    does next - is already contiguous and costs nothing."""
    lib = _lib.load()
    _check_inputs([("x", x), ("weight", norm.weight), ("bias", norm.bias)])
    _require(x.dtype == torch.float32 and x.dim() == 4 and x.shape[1] == norm.num_channels, "group_norm: fp32 NCHW")
    N, C, H, W = x.shape
    stats = torch.empty(N * norm.num_groups * 2, dtype=torch.float32, device=x.device)
    y = torch.empty((N, H * W, C) if tokens_out else (N, C, H, W), dtype=torch.float32, device=x.device)
    with _on(x.device):
        rc = lib.dfx_group_norm_f32(x.data_ptr(), norm.weight.data_ptr(), norm.bias.data_ptr(), stats.data_ptr(),
                                    y.data_ptr(), N, C, H * W, norm.num_groups, float(norm.eps), int(bool(tokens_out)),
                                    _stream(x.device))
    _lib.check(rc, "group_norm")
    return y.transpose(1, 2).unflatten(2, (H, W)) if tokens_out else y
