"""nn.MultiheadAttention forward (eval, no masks) on the hand-written kernels: joint input projections
on the MFMA GEMM, one fused attention launch (dfx_mha.h), output projection on the GEMM.

The library path spends ~12 launches per call on 300 x 300 problems (projection pieces, scaling, two
transposing copies, bmm, softmax, bmm, copy, projection); the 27 calls of a clip step are a fifth of its
launches.  Arithmetic is the module's own: q = (x_q W_q^T + b_q) / sqrt(d), softmax(q k^T) v, out_proj.
Inputs and output are batch-first [B, L, E] (the callers held batch-first tensors and transposed them
only for the module).
"""
import math

import torch

from dfx import ops as _ops

from .fused import apply_post, post_is_fusable


def usable(mha, *tensors):
    """Inference on the GPU in fp32 with 32-wide heads and a packed in_proj: the fused route applies."""
    return (not torch.is_grad_enabled() and mha.in_proj_weight is not None and mha.head_dim == 32
            and mha.in_proj_bias is not None and not mha.batch_first and mha.bias_k is None and not mha.add_zero_attn
            and all(t.is_cuda and t.dtype == torch.float32 for t in tensors))


def project_kv(mha, pool):
    """K and V projections of a POOL of key / value rows [rows, E] in one GEMM -> [rows, 2E] ([K | V]); rows gathered from it
    are the projections of the gathered rows (a Linear is row-wise), so a pool shared by many queries' key sets is projected once."""
    E = mha.embed_dim
    return _ops.linear(pool.contiguous(), mha.in_proj_weight[E:], mha.in_proj_bias[E:])


def project_kv_stack(mhas, pool):
    """``project_kv`` of several attention modules over the SAME pool in one launch (weights stacked along N, every result a
    contiguous [rows, 2E] tensor: dfx.ops.linear(col_block=2E)) - the three temporal query encoders of TransVOD++ pick from one
    pool of reference queries.  -> list of [rows, 2E]"""
    E = mhas[0].embed_dim
    key = tuple((m.in_proj_weight.data_ptr(), m.in_proj_weight._version, m.in_proj_bias._version) for m in mhas)
    cache = getattr(mhas[0], "_kv_stack", None)
    if cache is None or cache[0] != key:
        cache = (key, torch.cat([m.in_proj_weight[E:] for m in mhas], 0).contiguous(),
                 torch.cat([m.in_proj_bias[E:] for m in mhas], 0).contiguous())
        mhas[0]._kv_stack = cache
    out = _ops.linear(pool.contiguous(), cache[1], cache[2], col_block=2 * E)          # [n, rows, 2E]
    return [out[i] for i in range(len(mhas))]


def forward(mha, q_in, k_in, v_in, post=None, kv=None):
    """mha: nn.MultiheadAttention; q_in [B,Lq,E], k_in / v_in [B,Lk,E] -> [B,Lq,E]
    (= mha(q_in^T, k_in^T, v_in^T)[0]^T of the module; no caller uses the attention weights).
    post = (residual [B,Lq,E], norm[, dropout]): -> norm(residual + dropout(output)), the add and the LayerNorm in out_proj's
    GEMM epilogue when the dropout is the identity (eval mode).
    kv [B,Lk,2E]: already projected keys / values (``project_kv`` rows); k_in / v_in are then ignored."""
    E, H = mha.embed_dim, mha.num_heads
    W, b = mha.in_proj_weight, mha.in_proj_bias
    B, Lq, _ = q_in.shape
    if kv is not None:
        q = _ops.linear(q_in.contiguous(), W[:E], b[:E]).view(B, Lq, E)
        ctx = _ops.mha(q, kv[..., :E], kv[..., E:], H, 1.0 / math.sqrt(E // H))
        if post_is_fusable(post) and E == 256:
            return _ops.linear(ctx, mha.out_proj.weight, mha.out_proj.bias, residual=post[0].contiguous(), norm=post[1])
        out = _ops.linear(ctx, mha.out_proj.weight, mha.out_proj.bias)
        return apply_post(post, out)
    Lk = k_in.shape[1]
    same_qk, same_kv = q_in is k_in, k_in is v_in
    q_in, k_in, v_in = q_in.contiguous(), k_in.contiguous(), v_in.contiguous()
    if same_qk and same_kv:                                 # one projection for q, k, v
        qkv = _ops.linear(q_in, W, b).view(B, Lq, 3 * E)
        q, k, v = qkv[..., :E], qkv[..., E:2 * E], qkv[..., 2 * E:]
    elif same_qk:                                           # self-attention with positional q / k, plain v
        qk = _ops.linear(q_in, W[:2 * E], b[:2 * E]).view(B, Lq, 2 * E)
        q, k = qk[..., :E], qk[..., E:]
        v = _ops.linear(v_in, W[2 * E:], b[2 * E:]).view(B, Lk, E)
    else:
        q = _ops.linear(q_in, W[:E], b[:E]).view(B, Lq, E)
        k = _ops.linear(k_in, W[E:2 * E], b[E:2 * E]).view(B, Lk, E)
        v = _ops.linear(v_in, W[2 * E:], b[2 * E:]).view(B, Lk, E)
    ctx = _ops.mha(q, k, v, H, 1.0 / math.sqrt(E // H))
    if post_is_fusable(post) and E == 256:
        return _ops.linear(ctx, mha.out_proj.weight, mha.out_proj.bias, residual=post[0].contiguous(), norm=post[1])
    out = _ops.linear(ctx, mha.out_proj.weight, mha.out_proj.bias)
    return apply_post(post, out)
