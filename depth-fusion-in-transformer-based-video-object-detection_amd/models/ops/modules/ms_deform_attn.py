"""``MSDeformAttn``: the multi-scale deformable attention module, reference surface
(/root/reference/models/ops/modules/ms_deform_attn.py:30-117) on the gfx950 kernels.

Same constructor, parameter names (``sampling_offsets``, ``attention_weights``,
``value_proj``, ``output_proj`` -> same state_dict keys), initialisation and forward
signature as the reference.  Two execution routes:

* inference on the GPU with the production head geometry (8 heads x 32 channels, 4 points,
  <= 4 levels, fp32): ONE GEMM produces [offsets | logits] per query and the fused kernel
  (csrc/msda_fused.hip) does softmax, location arithmetic and sampling in one launch;
* anything else (training, fp64, odd geometry): the reference's op sequence with the
  autograd operator ``MSDeformAttnFunction`` (csrc/msda_forward.hip / msda_backward.hip).

There is no CPU route: like the reference's op (ms_deform_attn.h:38) it raises on CPU tensors.
"""
import math
import os
import warnings

import torch
import torch.nn.functional as F
from torch import nn

from models.fused import Linear, apply_post, post_is_fusable

from dfx import ops as _ops
from ..functions import ms_deform_attn_func as _func


def _is_power_of_2(n):
    if not isinstance(n, int) or n < 0:
        raise ValueError(f"invalid input for _is_power_of_2: {n} (type: {type(n)})")
    return n != 0 and (n & (n - 1)) == 0


class MSDeformAttn(nn.Module):
    def __init__(self, d_model=256, n_levels=4, n_heads=8, n_points=4):
        super().__init__()
        if d_model % n_heads != 0:
            raise ValueError(f"d_model must be divisible by n_heads, but got {d_model} and {n_heads}")
        if not _is_power_of_2(d_model // n_heads):
            warnings.warn("MSDeformAttn: a power-of-2 head dimension maps best onto the kernels "
                          "(32 channels per head takes the wave-per-query fast path).")
        self.im2col_step = 64
        self.d_model, self.n_levels, self.n_heads, self.n_points = d_model, n_levels, n_heads, n_points
        mlp = n_heads * n_levels * n_points
        self.sampling_offsets = Linear(d_model, mlp * 2)
        self.attention_weights = Linear(d_model, mlp)
        self.value_proj = Linear(d_model, d_model)
        self.output_proj = Linear(d_model, d_model)
        self._qproj_cache = None
        self._reset_parameters()

    def _reset_parameters(self):
        """ref :62-76 - zero offset weights, ring-shaped offset bias (head h points along angle
        2*pi*h/n_heads, point i at radius i+1), zero attention logits, xavier projections."""
        with torch.no_grad():
            self.sampling_offsets.weight.zero_()
            angle = torch.arange(self.n_heads, dtype=torch.float32) * (2.0 * math.pi / self.n_heads)
            ring = torch.stack([angle.cos(), angle.sin()], -1)
            ring = ring / ring.abs().max(-1, keepdim=True)[0]
            ring = ring.view(self.n_heads, 1, 1, 2).repeat(1, self.n_levels, self.n_points, 1)
            ring = ring * torch.arange(1, self.n_points + 1, dtype=torch.float32).view(1, 1, -1, 1)
            self.sampling_offsets.bias = nn.Parameter(ring.reshape(-1))
            self.attention_weights.weight.zero_()
            self.attention_weights.bias.zero_()
            nn.init.xavier_uniform_(self.value_proj.weight)
            self.value_proj.bias.zero_()
            nn.init.xavier_uniform_(self.output_proj.weight)
            self.output_proj.bias.zero_()

    # -- one GEMM for both query projections (weights concatenated lazily, refreshed when edited) --
    def _qproj_params(self):
        so, aw = self.sampling_offsets, self.attention_weights
        key = (so.weight._version, so.bias._version, aw.weight._version, aw.bias._version,
               so.weight.data_ptr(), aw.weight.data_ptr(), so.weight.device, so.weight.dtype)
        if self._qproj_cache is None or self._qproj_cache[0] != key:
            w = torch.cat([so.weight.detach(), aw.weight.detach()], 0).contiguous()
            b = torch.cat([so.bias.detach(), aw.bias.detach()], 0).contiguous()
            self._qproj_cache = (key, w, b)
        return self._qproj_cache[1], self._qproj_cache[2]

    def _qproj_params_by_head(self):
        """The joint projection with its output columns regrouped per head - [8 offsets | 4 logits] x n_heads
        (n_levels == 1) - so that a col_block=12 GEMM writes what one workgroup of the level-in-LDS kernel
        reads as one contiguous slab."""
        w, b = self._qproj_params()
        if self._qproj_cache[0] != getattr(self, "_qproj_head_key", None):
            M, P = self.n_heads, self.n_points
            so = torch.arange(M * P * 2, device=w.device).view(M, P * 2)
            aw = torch.arange(M * P, device=w.device).view(M, P) + M * P * 2
            order = torch.cat([so, aw], 1).reshape(-1)
            self._qproj_head = (w[order].contiguous(), b[order].contiguous())
            self._qproj_head_key = self._qproj_cache[0]
        return self._qproj_head

    def forward(self, query, reference_points, input_flatten, input_spatial_shapes, input_level_start_index,
                input_padding_mask=None, post=None, value=None):
        """query [N,Lq,C] (or a (tensor, positional_embedding) pair whose sum is the query: the add is
        then fused into the projection GEMM); reference_points [N,Lq,L,2] (or 4: cx,cy,w,h) in [0,1];
        input_flatten [N,sum(H_l*W_l),C]; input_spatial_shapes i64 [L,2]; input_level_start_index
        i64 [L]; input_padding_mask [N,S] True = padding.  -> [N,Lq,C]   (ref :78-117)
        post = (residual, norm[, dropout]): return ``norm(residual + dropout(output))`` instead - what every caller does
        next; on the fused GPU routes, and when the dropout is the identity (eval mode), the add and the LayerNorm ride in
        output_proj's GEMM epilogue (dfx.ops.linear(norm=...)).
        value [N,S,C]: ``value_proj(input_flatten)`` with the padded tokens already zeroed, when the caller projected the
        values of several layers that share ``input_flatten`` in one launch (``project_values``); the single-level
        many-query route (block-major operands) ignores it."""
        N, Lq, _ = (query[0] if isinstance(query, tuple) else query).shape
        _, S, _ = input_flatten.shape
        M, L, P = self.n_heads, self.n_levels, self.n_points
        D = self.d_model // M
        if getattr(input_spatial_shapes, "_dfx_tokens", None) != S:
            # one device read per distinct shapes tensor (the reference asserts on every call, ref :92)
            assert int((input_spatial_shapes[:, 0] * input_spatial_shapes[:, 1]).sum()) == S
            input_spatial_shapes._dfx_tokens = S
        ref_dim = reference_points.shape[-1]
        if ref_dim not in (2, 4):
            raise ValueError(f"Last dim of reference_points must be 2 or 4, but get {ref_dim} instead.")

        q_parts = query if isinstance(query, tuple) else (query,)
        no_grad = not torch.is_grad_enabled() or not (any(t.requires_grad for t in q_parts) or input_flatten.requires_grad)
        fused = no_grad and input_flatten.is_cuda and input_flatten.dtype == torch.float32 \
            and _ops.fused_supported(input_flatten, M, D, L, P, reference_points.shape[2])
        host = getattr(input_spatial_shapes, "_dfx_host", None)
        if fused and host is not None and _ops.level_supported(input_flatten, host[0][0], host[0][1], Lq, M, D, L, P,
                                                               reference_points.shape[2]):
            # single-level attention over many queries (encoder, depth fusion): the level lives in LDS and
            # the three projections exchange block-major operands with the kernel (include/dfx_msda.h)
            q, q_add = query if isinstance(query, tuple) else (query, None)
            w, b = self._qproj_params_by_head()
            value = _ops.linear(input_flatten.contiguous(), self.value_proj.weight, self.value_proj.bias,
                                row_mask=input_padding_mask, col_block=4)
            qproj = _ops.linear(q.contiguous(), w, b, add=None if q_add is None else q_add.contiguous(), col_block=12)
            sampled = _ops.msda_level_forward(value, reference_points, qproj, N, host[0][0], host[0][1])
            if post_is_fusable(post) and self.d_model == 256:
                return _ops.linear(sampled, self.output_proj.weight, self.output_proj.bias, x_blocked=True,
                                   residual=post[0].contiguous(), norm=post[1]).view(N, Lq, -1)
            out = _ops.linear(sampled, self.output_proj.weight, self.output_proj.bias, x_blocked=True).view(N, Lq, -1)
            return apply_post(post, out)
        if fused:
            # value_proj (+ masked_fill of padded tokens), [offsets | logits] in one GEMM (+ the
            # caller's ``src + pos`` add when handed over as a (src, pos) pair), fused sampling
            q, q_add = query if isinstance(query, tuple) else (query, None)
            w, b = self._qproj_params()
            if value is None:
                value = _ops.linear(input_flatten.contiguous(), self.value_proj.weight, self.value_proj.bias,
                                    row_mask=input_padding_mask)
            qproj = _ops.linear(q.contiguous(), w, b, add=None if q_add is None else q_add.contiguous())
            sampled = _ops.msda_fused_forward(value.view(N, S, M, D), input_spatial_shapes, input_level_start_index,
                                              reference_points, qproj, L, P)
            if post_is_fusable(post) and self.d_model == 256:
                return _ops.linear(sampled, self.output_proj.weight, self.output_proj.bias, residual=post[0].contiguous(),
                                   norm=post[1])
            out = self.output_proj(sampled)
            return apply_post(post, out)
        if isinstance(query, tuple):
            query = query[0] + query[1]

        if value is None:
            value = self.value_proj(input_flatten)
            if input_padding_mask is not None:
                value = value.masked_fill(input_padding_mask[..., None], float(0))
        value = value.view(N, S, M, D)

        offsets = self.sampling_offsets(query).view(N, Lq, M, L, P, 2)
        weights = F.softmax(self.attention_weights(query).view(N, Lq, M, L * P), -1).view(N, Lq, M, L, P)
        if ref_dim == 2:
            wh = torch.stack([input_spatial_shapes[..., 1], input_spatial_shapes[..., 0]], -1)
            locations = reference_points[:, :, None, :, None, :] + offsets / wh[None, None, None, :, None, :]
        else:
            locations = reference_points[:, :, None, :, None, :2] \
                + offsets / P * reference_points[:, :, None, :, None, 2:] * 0.5
        locations = locations.contiguous()
        if reference_points.shape[2] != L and N > 1:
            # flat read of an over-long location tensor (TransVOD temporal decoder): the
            # reference only does this with N == 1; each batch element follows that rule on its own
            sampled = torch.cat([_func.MSDeformAttnFunction.apply(
                value[b:b + 1].contiguous(), input_spatial_shapes, input_level_start_index,
                locations[b:b + 1].contiguous(), weights[b:b + 1].contiguous(), self.im2col_step)
                for b in range(N)], 0)
        else:
            sampled = _func.MSDeformAttnFunction.apply(value, input_spatial_shapes, input_level_start_index,
                                                       locations, weights, self.im2col_step)
        out = self.output_proj(sampled)
        return apply_post(post, out)


_STACK_VALUES = os.environ.get("DFX_VALUE_STACK", "1") == "1"          # 0: every layer projects its own values (A/B runs)


def project_values(attns, input_flatten, input_padding_mask=None):
    """``value_proj`` of several MSDeformAttn modules over the SAME ``input_flatten`` [N,S,C] in ONE GEMM launch (weights
    stacked along N, every result a contiguous [N,S,C] tensor of its own: dfx.ops.linear(col_block=C)) with the padded tokens
    zeroed - the six decoder layers of a frame read one memory, the three temporal decoders one current-frame memory
    (/root/reference/models/deformable_transformer_single.py:703-748, deformable_transformer_multi_plusplus.py:560-599 each
    project it on their own: the same products, the input read once instead of once per layer).
    -> list of value tensors for ``MSDeformAttn.forward(value=...)``, or None where the fused GPU route does not apply."""
    first = attns[0]
    C = first.d_model
    if (not _STACK_VALUES or len(attns) < 2 or torch.is_grad_enabled() or not input_flatten.is_cuda or input_flatten.dtype != torch.float32
            or C % 128 != 0 or any(a.d_model != C or a.value_proj.weight.dtype != torch.float32 or a.value_proj.bias is None
                                   or a.value_proj.weight.device != input_flatten.device for a in attns)):
        return None
    key = tuple((a.value_proj.weight.data_ptr(), a.value_proj.weight._version, a.value_proj.bias.data_ptr(), a.value_proj.bias._version)
                for a in attns)
    cache = getattr(first, "_value_stack", None)
    if cache is None or cache[0] != key:
        cache = (key, torch.cat([a.value_proj.weight for a in attns], 0).contiguous(),
                 torch.cat([a.value_proj.bias for a in attns], 0).contiguous())
        first._value_stack = cache
    N, S, _ = input_flatten.shape
    out = _ops.linear(input_flatten.contiguous(), cache[1], cache[2], row_mask=input_padding_mask, col_block=C)   # [n, N*S, C]
    return [out[i].view(N, S, C) for i in range(len(attns))]
