"""Transformer blocks that drive the MSDA operator, shared by the single-frame, TransVOD and
TransVOD++ transformers (the reference repeats them in each of its three transformer files;
parameter names below are the reference's, so checkpoints load unchanged).

Reference (all under /root/reference/models/):
  encoder layer / encoder        deformable_transformer_single.py:520-593
  decoder layer / decoder        deformable_transformer_single.py:596-748
  Late Fusion layer              deformable_transformer_single.py:341-402
  Encoder-CrossFusion layer/enc  deformable_transformer_single.py:406-518
  temporal query encoder layer   deformable_transformer_multi_plusplus.py:787-838
  temporal MSDA encoder layer    deformable_transformer_multi_plusplus.py:853-901
  temporal decoder               deformable_transformer_multi_plusplus.py:1030-1076
"""
import copy

import torch
import torch.nn.functional as F
from torch import nn

from models.fused import Linear, apply_post

from models.ops.modules import MSDeformAttn
from models.ops.modules.ms_deform_attn import project_values
from util.memo import memo_on
from util.misc import inverse_sigmoid


def _get_clones(module, n):
    return nn.ModuleList([copy.deepcopy(module) for _ in range(n)])


def _get_activation_fn(activation):
    try:
        return {"relu": F.relu, "gelu": F.gelu, "glu": F.glu}[activation]
    except KeyError:
        raise RuntimeError(f"activation should be relu/gelu, not {activation}.")


def _add_pos(x, pos):
    return x if pos is None else x + pos


def _linear_act(linear, activation, x):
    """activation(linear(x)); in GPU inference with ReLU the bias and the ReLU ride in the GEMM epilogue
    (one launch instead of two per FFN of the 300-query layers)."""
    if (activation is F.relu and x.is_cuda and x.dtype == torch.float32 and not torch.is_grad_enabled()
            and linear.in_features % 4 == 0):
        from dfx import ops as _ops
        return _ops.linear(x.contiguous(), linear.weight, linear.bias, relu=True)
    return activation(linear(x))


def _mha(module, q, k, v, post=None):
    """nn.MultiheadAttention on batch-first [B,L,E] tensors (no masks, attention weights unused): the fused
    GEMM + attention-kernel route in GPU inference (models/fused_mha.py), the module itself otherwise.
    post = (residual, norm, dropout): -> norm(residual + dropout(attention output)) (in out_proj's GEMM epilogue on the fused
    route when the dropout is the identity)."""
    from . import fused_mha
    if fused_mha.usable(module, q, k, v):
        return fused_mha.forward(module, q, k, v, post)
    out = module(q.transpose(0, 1), k.transpose(0, 1), v.transpose(0, 1))[0].transpose(0, 1)
    return apply_post(post, out)


def _gpu_inference(x):
    return x.is_cuda and x.dtype == torch.float32 and not torch.is_grad_enabled()


def _identity_dropout(dropout):
    return dropout is None or not dropout.training or dropout.p == 0


def _linear_norm_add(linear, x, norm, residual=None, act=None, act_first=False, dropout=None):
    """norm(residual + dropout(act(linear(x)))) (act_first) or norm(act(linear(x) + residual)); one MFMA GEMM launch with the
    residual add and the LayerNorm in its epilogue in GPU inference when the Linear ends in d_model = 256 columns
    (dfx.ops.linear(norm=...)) and the sub-layer's Dropout is the identity (eval mode), the separate ops otherwise."""
    if (_gpu_inference(x) and linear.out_features == 256 and linear.in_features % 4 == 0 and linear.weight.dtype == torch.float32
            and _identity_dropout(dropout)):
        from dfx import ops as _ops
        return _ops.linear(x.contiguous(), linear.weight, linear.bias, act=act, residual=None if residual is None else residual.contiguous(),
                           norm=norm, act_first=act_first)
    y = linear(x)
    fn = {None: None, "relu": F.relu, "gelu": F.gelu}[act]
    if fn is not None and act_first:
        y = fn(y)
    if dropout is not None:
        y = dropout(y)
    if residual is not None:
        y = residual + y
    if fn is not None and not act_first:
        y = fn(y)
    return norm(y)


def _norm_add(norm, x, y=None):
    """norm(x + y).  Inference on the GPU: one fused pass (dfx.ops.add_layernorm); otherwise the
    reference's two ops."""
    if x.is_cuda and x.dtype == torch.float32 and not torch.is_grad_enabled() and x.shape[-1] % 4 == 0 \
            and x.shape[-1] <= 1024:
        from dfx import ops as _ops
        return _ops.add_layernorm(x, y, norm)
    return norm(x if y is None else x + y)


# ---- level bookkeeping: device tensors for the kernels, host copies for Python loops -------------
_LEVEL_CACHE = {}


def make_level_tensors(shape_list, device):
    """(spatial_shapes i64 [L,2], level_start_index i64 [L]) on ``device`` from host (H,W) pairs.
    The host list rides along (``_dfx_host``) so later code never reads sizes back from the GPU, and
    the pair is cached per (shapes, device): building it is a blocking host-to-device copy, i.e. a
    full stream synchronisation per forward if done every time."""
    shape_list = [(int(h), int(w)) for h, w in shape_list]
    key = (tuple(shape_list), str(device))
    hit = _LEVEL_CACHE.get(key)
    if hit is not None:
        return hit
    starts, acc = [], 0
    for h, w in shape_list:
        starts.append(acc)
        acc += h * w
    shapes = torch.as_tensor(shape_list, dtype=torch.long, device=device)
    lsi = torch.as_tensor(starts, dtype=torch.long, device=device)
    shapes._dfx_host = shape_list
    shapes._dfx_tokens = acc
    if len(_LEVEL_CACHE) > 64:
        _LEVEL_CACHE.clear()
    _LEVEL_CACHE[key] = (shapes, lsi)
    return shapes, lsi


def host_shapes(spatial_shapes):
    host = getattr(spatial_shapes, "_dfx_host", None)
    return host if host is not None else [(int(h), int(w)) for h, w in spatial_shapes.tolist()]


def get_valid_ratio(mask):
    """Fraction of each padded map that is image: [N,2] = (w_ratio, h_ratio)."""
    return memo_on(mask, "valid_ratio", lambda: _valid_ratio(mask))


def _valid_ratio(mask):
    _, H, W = mask.shape
    valid_h = torch.sum(~mask[:, :, 0], 1)
    valid_w = torch.sum(~mask[:, 0, :], 1)
    return torch.stack([valid_w.float() / W, valid_h.float() / H], -1)


def get_reference_points(spatial_shapes, valid_ratios, device):
    """Pixel-centre grid of every level in valid-image coordinates, [N, sum(HW), L, 2]
    (ref deformable_transformer_single.py:165-177)."""
    per_level = []
    for lvl, (H, W) in enumerate(host_shapes(spatial_shapes)):
        ys = torch.linspace(0.5, H - 0.5, H, dtype=torch.float32, device=device)
        xs = torch.linspace(0.5, W - 0.5, W, dtype=torch.float32, device=device)
        gy, gx = torch.meshgrid(ys, xs, indexing="ij")
        gy = gy.reshape(-1)[None] / (valid_ratios[:, None, lvl, 1] * H)
        gx = gx.reshape(-1)[None] / (valid_ratios[:, None, lvl, 0] * W)
        per_level.append(torch.stack((gx, gy), -1))
    grid = torch.cat(per_level, 1)
    return grid[:, :, None] * valid_ratios[:, None]


# ---- encoder ----------------------------------------------------------------------------------------
class DeformableTransformerEncoderLayer(nn.Module):
    def __init__(self, d_model=256, d_ffn=1024, dropout=0.1, activation="relu", n_levels=4, n_heads=8, n_points=4):
        super().__init__()
        self.self_attn = MSDeformAttn(d_model, n_levels, n_heads, n_points)
        self.dropout1 = nn.Dropout(dropout)
        self.norm1 = nn.LayerNorm(d_model)
        self.linear1 = Linear(d_model, d_ffn)
        self.activation = _get_activation_fn(activation)
        self.dropout2 = nn.Dropout(dropout)
        self.linear2 = Linear(d_ffn, d_model)
        self.dropout3 = nn.Dropout(dropout)
        self.norm2 = nn.LayerNorm(d_model)

    with_pos_embed = staticmethod(_add_pos)

    def forward_ffn(self, src):
        y = self.linear2(self.dropout2(self.activation(self.linear1(src))))
        return _norm_add(self.norm2, src, self.dropout3(y))

    def forward(self, src, pos, reference_points, spatial_shapes, level_start_index, padding_mask=None,
                rgbd_src=None):
        fused = src.is_cuda and not torch.is_grad_enabled() and src.dtype == torch.float32
        if rgbd_src is not None:
            query = rgbd_src
        elif fused and pos is not None:
            query = (src, pos)                     # the add rides along into the projection GEMM
        else:
            query = _add_pos(src, pos)
        if fused:       # residual add + LayerNorm in output_proj's epilogue
            src = self.self_attn(query, reference_points, src, spatial_shapes, level_start_index, padding_mask,
                                 post=(src, self.norm1, self.dropout1))
        else:
            y = self.self_attn(query, reference_points, src, spatial_shapes, level_start_index, padding_mask)
            src = _norm_add(self.norm1, src, self.dropout1(y))
        if fused and self.activation is F.relu:
            from dfx import ops as _ops            # linear1 + bias + ReLU in one MFMA GEMM, linear2 + residual + LayerNorm in another
            h = _ops.linear(src.contiguous(), self.linear1.weight, self.linear1.bias, relu=True)
            return _linear_norm_add(self.linear2, self.dropout2(h), self.norm2, src, dropout=self.dropout3)
        return self.forward_ffn(src)


class DeformableTransformerEncoder(nn.Module):
    def __init__(self, encoder_layer, num_layers):
        super().__init__()
        self.layers = _get_clones(encoder_layer, num_layers)
        self.num_layers = num_layers

    get_reference_points = staticmethod(get_reference_points)

    def forward(self, src, spatial_shapes, level_start_index, valid_ratios, pos=None, padding_mask=None,
                rgbd_src=None):
        out = src
        ref = get_reference_points(spatial_shapes, valid_ratios, device=src.device)
        for layer in self.layers:
            out = layer(out, pos, ref, spatial_shapes, level_start_index, padding_mask, rgbd_src=rgbd_src)
        return out


# ---- depth fusion blocks ----------------------------------------------------------------------------
class _CrossFusionBlock(nn.Module):
    """RGB tokens attend to depth tokens with MSDA:
         src  = LN(Linear(depth))
         tgt2 = Linear(MSDA(q = rgb + pos, ref = rgb grid, value = src))
         tgt  = LN(tgt + tgt2);  tgt = LN(tgt + GELU(Linear(tgt)))
    Late Fusion and Encoder Cross Fusion differ only in the name of the last LayerNorm."""

    _ffn_norm = "norm3"
    _ffn_drop = "dropout4"

    def __init__(self, d_model, dropout, n_levels, n_heads, n_points):
        super().__init__()
        self.cross_attn = MSDeformAttn(d_model, n_levels, n_heads, n_points)
        self.dropout1 = nn.Dropout(dropout)
        self.norm1 = nn.LayerNorm(d_model)
        self.linear1 = Linear(d_model, d_model)
        self.activation = _get_activation_fn("gelu")
        setattr(self, self._ffn_drop, nn.Dropout(dropout))
        setattr(self, self._ffn_norm, nn.LayerNorm(d_model))
        self.depth_scale_adapt = Linear(d_model, d_model)
        self.norm_depth_scale = nn.LayerNorm(d_model)
        self.cross_scale_adapt = Linear(d_model, d_model)

    with_pos_embed = staticmethod(_add_pos)

    def forward_ffn(self, tgt):
        if (self.activation is F.gelu and tgt.is_cuda and tgt.dtype == torch.float32 and not torch.is_grad_enabled()):
            from dfx import ops as _ops            # Linear + bias + exact GELU in one MFMA GEMM
            # Linear + bias + exact GELU + residual + LayerNorm in one MFMA GEMM
            return _linear_norm_add(self.linear1, tgt, getattr(self, self._ffn_norm), tgt, act="gelu", act_first=True,
                                    dropout=getattr(self, self._ffn_drop))
        y = self.activation(self.linear1(tgt))
        return _norm_add(getattr(self, self._ffn_norm), tgt, getattr(self, self._ffn_drop)(y))

    def _fuse(self, tgt, query_pos, reference_points, src, src_spatial_shapes, src_start_index, src_padding_mask):
        src = _linear_norm_add(self.depth_scale_adapt, src, self.norm_depth_scale)
        fused = tgt.is_cuda and not torch.is_grad_enabled() and tgt.dtype == torch.float32 and query_pos is not None
        query = (tgt, query_pos) if fused else _add_pos(tgt, query_pos)
        y = self.cross_attn(query, reference_points, src, src_spatial_shapes, src_start_index, src_padding_mask)
        tgt = _linear_norm_add(self.cross_scale_adapt, y, self.norm1, tgt, dropout=self.dropout1)
        return self.forward_ffn(tgt)


class DepthDeformableTransformerEncoderLayer(_CrossFusionBlock):
    """Late Fusion layer (ref deformable_transformer_single.py:341-402)."""

    def __init__(self, d_model=256, d_ffn=1024, dropout=0.1, activation="relu", n_depth_levels=1, n_heads=8,
                 dpth_n_points=4, depth_self_attn=False, gate=True, adaptation_layers=True):
        super().__init__(d_model, dropout, n_depth_levels, n_heads, dpth_n_points)
        self.depth_self_attn = depth_self_attn
        self.adaptation_layers = adaptation_layers

    def forward(self, tgt, query_pos, src_pos, tgt_spatial_shapes, reference_points, depth_reference_points, src,
                src_spatial_shapes, frame_start_index, tgt_padding_mask=None, src_padding_mask=None):
        return self._fuse(tgt, query_pos, reference_points, src, src_spatial_shapes, frame_start_index,
                          src_padding_mask)


class DeformableTransformerFusionLayerV2(_CrossFusionBlock):
    """Encoder Cross Fusion layer (ref deformable_transformer_single.py:406-461)."""

    _ffn_norm = "norm2"
    _ffn_drop = "dropout3"

    def __init__(self, d_model=256, d_ffn=1024, dropout=0.1, activation="gelu", n_levels=4, n_heads=8, n_points=4):
        super().__init__(d_model, dropout, n_levels, n_heads, n_points)

    def forward(self, tgt, query_pos, reference_points, src, src_spatial_shapes, level_start_index,
                src_padding_mask=None):
        return self._fuse(tgt, query_pos, reference_points, src, src_spatial_shapes, level_start_index,
                          src_padding_mask)


class RGBDDeformableTransformerEncoderV2(nn.Module):
    """RGB encoder with a fusion layer after the first ``depth_num_layers`` RGB layers; the fusion
    layer's value is the previous fusion output (depth tokens at first) and its result is added to
    the RGB stream (ref deformable_transformer_single.py:465-518)."""

    def __init__(self, encoder_layer, fusion_encoder_layer, num_layers, depth_num_layers, fusion_num_layers,
                 fusion_layers_order=[]):
        super().__init__()
        self.layers = _get_clones(encoder_layer, num_layers)
        self.fusion_layers = _get_clones(fusion_encoder_layer, fusion_num_layers)
        self.num_layers = num_layers
        self.depth_num_layers = depth_num_layers
        self.fusion_num_layers = fusion_num_layers
        self.fusion_layers_order = list(fusion_layers_order) if len(fusion_layers_order) > 0 \
            else list(range(fusion_num_layers))
        assert len(self.fusion_layers_order) == self.fusion_num_layers, \
            "The number of fusion layers should match the fusion layer count"

    get_reference_points = staticmethod(get_reference_points)

    def forward(self, src, spatial_shapes, level_start_index, valid_ratios, pos=None, padding_mask=None,
                rgbd_src=None, depth_src=None, depth_spatial_shapes=None, depth_level_start_index=None,
                depth_valid_ratios=None, depth_pos=None, depth_padding_mask=None):
        out, fused = src, depth_src
        ref = get_reference_points(spatial_shapes, valid_ratios, device=src.device)
        for i, layer in enumerate(self.layers):
            out = layer(out, pos, ref, spatial_shapes, level_start_index, padding_mask)
            if i < self.depth_num_layers and i in self.fusion_layers_order:
                fusion = self.fusion_layers[self.fusion_layers_order.index(i)]
                fused = fusion(out, pos, ref, fused, depth_spatial_shapes, depth_level_start_index, padding_mask)
                out = out + fused
        return out


# ---- decoder ----------------------------------------------------------------------------------------
class DeformableTransformerDecoderLayer(nn.Module):
    def __init__(self, d_model=256, d_ffn=1024, dropout=0.1, activation="relu", n_levels=4, n_heads=8, n_points=4):
        super().__init__()
        self.cross_attn = MSDeformAttn(d_model, n_levels, n_heads, n_points)
        self.dropout1 = nn.Dropout(dropout)
        self.norm1 = nn.LayerNorm(d_model)
        self.self_attn = nn.MultiheadAttention(d_model, n_heads, dropout=dropout)
        self.dropout2 = nn.Dropout(dropout)
        self.norm2 = nn.LayerNorm(d_model)
        self.linear1 = Linear(d_model, d_ffn)
        self.activation = _get_activation_fn(activation)
        self.dropout3 = nn.Dropout(dropout)
        self.linear2 = Linear(d_ffn, d_model)
        self.dropout4 = nn.Dropout(dropout)
        self.norm3 = nn.LayerNorm(d_model)

    with_pos_embed = staticmethod(_add_pos)

    def forward_ffn(self, tgt):
        return _linear_norm_add(self.linear2, self.dropout3(_linear_act(self.linear1, self.activation, tgt)), self.norm3, tgt,
                                dropout=self.dropout4)

    def forward(self, tgt, query_pos, reference_points, src, src_spatial_shapes, level_start_index,
                src_padding_mask=None, value=None):
        """value: this layer's ``cross_attn.value_proj(src)`` when the decoder projected all its layers' values in one
        launch (``project_values``)."""
        qk = _add_pos(tgt, query_pos)
        tgt = _mha(self.self_attn, qk, qk, tgt, post=(tgt, self.norm2, self.dropout2))
        tgt = self.cross_attn(_add_pos(tgt, query_pos), reference_points, src, src_spatial_shapes,
                              level_start_index, src_padding_mask, post=(tgt, self.norm1, self.dropout1), value=value)
        return self.forward_ffn(tgt)


def _scale_reference(reference_points, valid_ratios):
    if reference_points.shape[-1] == 4:
        vr4 = memo_on(valid_ratios, "vr4", lambda: torch.cat([valid_ratios, valid_ratios], -1)[:, None])
        return reference_points[:, :, None] * vr4
    assert reference_points.shape[-1] == 2
    return reference_points[:, :, None] * valid_ratios[:, None]


class DeformableTransformerDecoder(nn.Module):
    """Stack of decoder layers with optional per-layer box refinement: after layer i the reference
    boxes become sigmoid(bbox_embed[i](out) + inverse_sigmoid(ref)), detached
    (ref deformable_transformer_single.py:703-748)."""

    def __init__(self, decoder_layer, num_layers, return_intermediate=False):
        super().__init__()
        self.layers = _get_clones(decoder_layer, num_layers)
        self.num_layers = num_layers
        self.return_intermediate = return_intermediate
        self.bbox_embed = None    # set by the detector (box refinement / two-stage)
        self.class_embed = None

    def _refine(self, lid, output, reference_points):
        if self.bbox_embed is None:
            return reference_points
        delta = self.bbox_embed[lid](output)
        if delta.is_cuda and delta.dtype == torch.float32 and not torch.is_grad_enabled():
            from dfx import ops as _ops            # sigmoid(delta + inverse_sigmoid(ref)) in one launch
            return _ops.box_refine(delta, reference_points)
        if reference_points.shape[-1] == 4:
            new = delta + inverse_sigmoid(reference_points)
        else:
            assert reference_points.shape[-1] == 2
            new = delta
            new[..., :2] = delta[..., :2] + inverse_sigmoid(reference_points)
        return new.sigmoid().detach()

    def forward(self, tgt, reference_points, src, src_spatial_shapes, src_level_start_index, src_valid_ratios,
                query_pos=None, src_padding_mask=None, values=None):
        """values: the layers' value projections of ``src`` when the caller already has them (the three temporal decoders of
        TransVOD++ share one memory); else all layers' are projected here in one launch (GPU inference)."""
        output = tgt
        inter, inter_refs = [], []
        if values is None and _gpu_inference(src):
            values = project_values([layer.cross_attn for layer in self.layers], src, src_padding_mask)
        for lid, layer in enumerate(self.layers):
            ref_in = _scale_reference(reference_points, src_valid_ratios)
            output = layer(output, query_pos, ref_in, src, src_spatial_shapes, src_level_start_index,
                           src_padding_mask, **({} if values is None else {"value": values[lid]}))
            reference_points = self._refine(lid, output, reference_points)
            if self.return_intermediate:
                inter.append(output)
                inter_refs.append(reference_points)
        if self.return_intermediate:
            return torch.stack(inter), torch.stack(inter_refs)
        return output, reference_points


class TemporalDeformableTransformerDecoder(DeformableTransformerDecoder):
    """The TDTD decoder of TransVOD++: same layers, but box refinement is switched off on every
    call (the reference resets ``self.bbox_embed = None`` inside the loop,
    deformable_transformer_multi_plusplus.py:1055)."""

    def _refine(self, lid, output, reference_points):
        self.bbox_embed = None
        return reference_points


# ---- temporal query / memory layers -----------------------------------------------------------------
class TemporalQueryEncoderLayer(nn.Module):
    """TQE: self-attention over the current frame's queries, then cross-attention to the selected
    reference-frame queries, then FFN (ref deformable_transformer_multi_plusplus.py:787-838)."""

    def __init__(self, d_model=256, d_ffn=1024, dropout=0.1, activation="relu", n_heads=8):
        super().__init__()
        self.self_attn = nn.MultiheadAttention(d_model, n_heads, dropout=dropout)
        self.dropout2 = nn.Dropout(dropout)
        self.norm2 = nn.LayerNorm(d_model)
        self.cross_attn = nn.MultiheadAttention(d_model, n_heads, dropout=dropout)
        self.dropout1 = nn.Dropout(dropout)
        self.norm1 = nn.LayerNorm(d_model)
        self.linear1 = Linear(d_model, d_ffn)
        self.activation = _get_activation_fn(activation)
        self.dropout3 = nn.Dropout(dropout)
        self.linear2 = Linear(d_ffn, d_model)
        self.dropout4 = nn.Dropout(dropout)
        self.norm3 = nn.LayerNorm(d_model)

    with_pos_embed = staticmethod(_add_pos)

    def forward_ffn(self, tgt):
        return _linear_norm_add(self.linear2, self.dropout3(_linear_act(self.linear1, self.activation, tgt)), self.norm3, tgt,
                                dropout=self.dropout4)

    def forward(self, query, ref_query, query_pos=None, ref_query_pos=None, ref_kv=None):
        """ref_kv [B,Lk,2E]: the cross-attention's key / value projections of ``ref_query``, when the caller gathered them from
        a pool it projected once (``project_ref_pool``; GPU inference route); ``ref_query`` is then not read."""
        qk = _add_pos(query, query_pos)
        tgt = _mha(self.self_attn, qk, qk, query, post=(query, self.norm2, self.dropout2))
        if ref_kv is not None:
            from . import fused_mha
            tgt = fused_mha.forward(self.cross_attn, _add_pos(tgt, query_pos), None, None, post=(tgt, self.norm1, self.dropout1), kv=ref_kv)
        else:
            tgt = _mha(self.cross_attn, _add_pos(tgt, query_pos), _add_pos(ref_query, ref_query_pos), ref_query,
                       post=(tgt, self.norm1, self.dropout1))
        return self.forward_ffn(tgt)

    def project_ref_pool(self, pool):
        """[rows, E] reference queries -> [rows, 2E] their key / value projections for ``forward(ref_kv=...)``, or None when
        the fused GPU route does not apply (the caller then hands the gathered queries over as before)."""
        from . import fused_mha
        if fused_mha.usable(self.cross_attn, pool):
            return fused_mha.project_kv(self.cross_attn, pool)
        return None


class TemporalQueryEncoder(nn.Module):
    def __init__(self, encoder_layer, num_layers):
        super().__init__()
        self.layers = _get_clones(encoder_layer, num_layers)
        self.num_layers = num_layers

    def forward(self, query, ref_query, query_pos=None, ref_query_pos=None):
        out = query
        for layer in self.layers:
            out = layer(out, ref_query, query_pos, ref_query_pos)
        return out


class TemporalDeformableTransformerEncoderLayer(nn.Module):
    """TDAM memory layer of TransVOD: the current frame attends to the reference frames' memories
    with MSDA, one "level" per reference frame (ref deformable_transformer_multi_plusplus.py:853-901)."""

    def __init__(self, d_model=256, d_ffn=1024, dropout=0.1, activation="relu", num_ref_frames=3, n_heads=8,
                 n_points=4):
        super().__init__()
        self.cross_attn = MSDeformAttn(d_model, num_ref_frames, n_heads, n_points)
        self.dropout1 = nn.Dropout(dropout)
        self.norm1 = nn.LayerNorm(d_model)
        self.self_attn = nn.MultiheadAttention(d_model, n_heads, dropout=dropout)
        self.dropout2 = nn.Dropout(dropout)
        self.norm2 = nn.LayerNorm(d_model)
        self.linear1 = Linear(d_model, d_ffn)
        self.activation = _get_activation_fn(activation)
        self.dropout3 = nn.Dropout(dropout)
        self.linear2 = Linear(d_ffn, d_model)
        self.dropout4 = nn.Dropout(dropout)
        self.norm3 = nn.LayerNorm(d_model)

    with_pos_embed = staticmethod(_add_pos)

    def forward_ffn(self, tgt):
        return _linear_norm_add(self.linear2, self.dropout3(_linear_act(self.linear1, self.activation, tgt)), self.norm3, tgt,
                                dropout=self.dropout4)

    def forward(self, tgt, query_pos, reference_points, src, src_spatial_shapes, frame_start_index,
                src_padding_mask=None):
        qk = _add_pos(tgt, query_pos)
        tgt = _mha(self.self_attn, qk, qk, tgt, post=(tgt, self.norm2, self.dropout2))
        tgt = self.cross_attn(_add_pos(tgt, query_pos), reference_points, src, src_spatial_shapes,
                              frame_start_index, src_padding_mask, post=(tgt, self.norm1, self.dropout1))
        return self.forward_ffn(tgt)
