"""Positional encodings fed to the attention queries (ref models/position_encoding.py:20-97)."""
import math

import torch
from torch import nn

from util.memo import memo_on
from util.misc import NestedTensor


class PositionEmbeddingSine(nn.Module):
    """Sin/cos of the (cumulative, optionally normalised) row and column index of every valid
    pixel; ``num_pos_feats`` channels for y followed by ``num_pos_feats`` for x (ref :36-56)."""

    def __init__(self, num_pos_feats=64, temperature=10000, normalize=False, scale=None):
        super().__init__()
        if scale is not None and normalize is False:
            raise ValueError("normalize should be True if scale is passed")
        self.num_pos_feats, self.temperature, self.normalize = num_pos_feats, temperature, normalize
        self.scale = 2 * math.pi if scale is None else scale

    def forward(self, tensor_list: NestedTensor):
        mask = tensor_list.mask
        assert mask is not None
        return memo_on(mask, ("pos_sine", self.num_pos_feats, self.temperature, self.normalize, self.scale),
                       lambda: self._encode(mask))

    def _encode(self, mask):
        valid = ~mask
        ys = valid.cumsum(1, dtype=torch.float32)
        xs = valid.cumsum(2, dtype=torch.float32)
        if self.normalize:
            eps = 1e-6
            ys = (ys - 0.5) / (ys[:, -1:, :] + eps) * self.scale
            xs = (xs - 0.5) / (xs[:, :, -1:] + eps) * self.scale
        k = torch.arange(self.num_pos_feats, dtype=torch.float32, device=mask.device)
        freq = self.temperature ** (2 * (k // 2) / self.num_pos_feats)

        def encode(v):
            a = v[:, :, :, None] / freq
            return torch.stack((a[:, :, :, 0::2].sin(), a[:, :, :, 1::2].cos()), dim=4).flatten(3)

        return torch.cat((encode(ys), encode(xs)), dim=3).permute(0, 3, 1, 2)


class PositionEmbeddingLearned(nn.Module):
    """Learned row/column tables of 50 entries each (ref :59-83)."""

    def __init__(self, num_pos_feats=256):
        super().__init__()
        self.row_embed = nn.Embedding(50, num_pos_feats)
        self.col_embed = nn.Embedding(50, num_pos_feats)
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.uniform_(self.row_embed.weight)
        nn.init.uniform_(self.col_embed.weight)

    def forward(self, tensor_list: NestedTensor):
        x = tensor_list.tensors
        h, w = x.shape[-2:]
        col = self.col_embed(torch.arange(w, device=x.device))
        row = self.row_embed(torch.arange(h, device=x.device))
        pos = torch.cat([col.unsqueeze(0).expand(h, -1, -1), row.unsqueeze(1).expand(-1, w, -1)], dim=-1)
        return pos.permute(2, 0, 1).unsqueeze(0).repeat(x.shape[0], 1, 1, 1)


def build_position_encoding(args):
    n = args.hidden_dim // 2
    if args.position_embedding in ("v2", "sine"):
        return PositionEmbeddingSine(n, normalize=True)
    if args.position_embedding in ("v3", "learned"):
        return PositionEmbeddingLearned(n)
    raise ValueError(f"not supported {args.position_embedding}")
