"""Query / RoI fusion head of TransVOD++ (Sparse R-CNN style), ref models/sparse_roi_head/head.py.

``RCNNHead`` (:31-83): self-attention over the N*300 object queries, ``DynamicConv`` interaction
with each query's 7x7 RoI feature, FFN.  ``DynamicConv`` (:127-172): every query generates two
1x1 "conv" kernels (256x64 and 64x256) with one Linear 256 -> 2*256*64, applies them to its own
49x256 RoI feature as two batched matmuls with LayerNorm+ReLU, flattens 49*256 and projects back
to 256.
"""
import math

import torch
import torch.nn.functional as F
from torch import nn

from models.fused import Linear
from models.transformer_layers import _linear_norm_add, _norm_add

_DEFAULT_SCALE_CLAMP = math.log(100000.0 / 16)


def _get_activation_fn(activation):
    try:
        return {"relu": F.relu, "gelu": F.gelu, "glu": F.glu}[activation]
    except KeyError:
        raise RuntimeError(f"activation should be relu/gelu, not {activation}.")


class DynamicConv(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        sp = cfg["MODEL"]["SparseRCNN"]
        self.hidden_dim, self.dim_dynamic, self.num_dynamic = sp["HIDDEN_DIM"], sp["DIM_DYNAMIC"], sp["NUM_DYNAMIC"]
        self.num_params = self.hidden_dim * self.dim_dynamic
        self.dynamic_layer = Linear(self.hidden_dim, self.num_dynamic * self.num_params)
        self.norm1 = nn.LayerNorm(self.dim_dynamic)
        self.norm2 = nn.LayerNorm(self.hidden_dim)
        self.activation = nn.ReLU(inplace=True)
        res = cfg["MODEL"]["ROI_BOX_HEAD"]["POOLER_RESOLUTION"]
        self.out_layer = Linear(self.hidden_dim * res ** 2, self.hidden_dim)
        self.norm3 = nn.LayerNorm(self.hidden_dim)

    def forward(self, pro_features, roi_features, params=None):
        """pro_features (1, K, C); roi_features (49, K, C) -> (K, C).  ``params``: the output of ``dynamic_layer(pro_features)``
        when the caller already has it (it depends on the queries only: RCNNHead shares it between RoI feature sets)."""
        feats = roi_features.permute(1, 0, 2)                               # K,49,C
        if params is None:
            params = self.dynamic_layer(pro_features)
        if (feats.is_cuda and feats.dtype == torch.float32 and not torch.is_grad_enabled() and feats.is_contiguous()
                and self.hidden_dim == 256 and self.dim_dynamic == 64 and feats.shape[1] <= 64
                and isinstance(self.activation, nn.ReLU)):
            # both per-RoI products and their LayerNorm + ReLU in one launch (csrc/dynconv.hip)
            from dfx import ops as _ops
            params2 = params.view(feats.shape[0], -1)                                       # K, 2*C*dd
            feats = _ops.dynamic_conv(feats, params2, self.norm1, self.norm2)
            feats = self.out_layer(feats.flatten(1))
            return self.activation(_norm_add(self.norm3, feats))
        params = params.permute(1, 0, 2)                                    # K,1,2*C*dd
        k1 = params[:, :, : self.num_params].reshape(-1, self.hidden_dim, self.dim_dynamic)
        k2 = params[:, :, self.num_params:].reshape(-1, self.dim_dynamic, self.hidden_dim)
        feats = self.activation(self.norm1(torch.bmm(feats, k1)))
        feats = self.activation(self.norm2(torch.bmm(feats, k2)))
        feats = self.out_layer(feats.flatten(1))
        return self.activation(self.norm3(feats))


class RCNNHead(nn.Module):
    def __init__(self, cfg, d_model, num_classes, dim_feedforward=2048, nhead=8, dropout=0.1, activation="relu",
                 scale_clamp: float = _DEFAULT_SCALE_CLAMP, bbox_weights=(2.0, 2.0, 1.0, 1.0)):
        super().__init__()
        self.d_model = d_model
        self.self_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.inst_interact = DynamicConv(cfg)
        self.linear1 = Linear(d_model, dim_feedforward)
        self.dropout = nn.Dropout(dropout)
        self.linear2 = Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.norm3 = nn.LayerNorm(d_model)
        self.dropout1 = nn.Dropout(dropout)
        self.dropout2 = nn.Dropout(dropout)
        self.dropout3 = nn.Dropout(dropout)
        self.activation = _get_activation_fn(activation)
        self.scale_clamp, self.bbox_weights = scale_clamp, bbox_weights

    def forward(self, roi_features, pro_features):
        """roi_features (N*nr_boxes, C, 7, 7); pro_features (N, nr_boxes, C) -> (1, N*nr_boxes, C).
        A list / tuple of RoI feature tensors (the same boxes pooled from several maps: TransVOD++ pools every frame's boxes
        from its plain memory and from memory + positions) gives a list of outputs; the self-attention over the queries and the
        32768-wide ``dynamic_layer`` - which see the queries only - then run once instead of once per feature set."""
        if isinstance(roi_features, (list, tuple)):
            q = self._queries(pro_features)
            params = self.inst_interact.dynamic_layer(q)
            return [self._interact(q, r, pro_features, params) for r in roi_features]
        return self._interact(self._queries(pro_features), roi_features, pro_features)

    def _queries(self, pro_features):
        N, nr_boxes = pro_features.shape[:2]
        q = pro_features.view(N, nr_boxes, self.d_model).permute(1, 0, 2)                 # nr,N,C
        from .. import fused_mha
        if fused_mha.usable(self.self_attn, pro_features):
            # batch-first all the way: residual + LayerNorm ride in out_proj's GEMM epilogue
            pf = pro_features.reshape(N, nr_boxes, self.d_model)
            q = fused_mha.forward(self.self_attn, pf, pf, pf, post=(pf, self.norm1, self.dropout1)).reshape(1, N * nr_boxes, self.d_model)
        else:
            attn = self.self_attn(q, q, value=q)[0]
            q = _norm_add(self.norm1, q, self.dropout1(attn))
            q = q.view(nr_boxes, N, self.d_model).permute(1, 0, 2).reshape(1, N * nr_boxes, self.d_model)
        return q

    def _interact(self, q, roi_features, pro_features, params=None):
        N, nr_boxes = pro_features.shape[:2]
        roi = roi_features.view(N * nr_boxes, self.d_model, -1).permute(2, 0, 1)          # 49,K,C
        obj = _norm_add(self.norm2, q, self.dropout2(self.inst_interact(q, roi, params)).view_as(q))
        if (self.activation is F.relu and obj.is_cuda and obj.dtype == torch.float32 and not torch.is_grad_enabled()):
            from dfx import ops as _ops            # bias + ReLU in the GEMM epilogue
            hdn = _ops.linear(obj.contiguous(), self.linear1.weight, self.linear1.bias, relu=True)
        else:
            hdn = self.activation(self.linear1(obj))
        return _linear_norm_add(self.linear2, self.dropout(hdn), self.norm3, obj, dropout=self.dropout3)
