"""Inference-time memo of small tensors derived from a (mask) tensor.

The padding mask of a clip does not change between calls, but everything derived from it - the
resized masks, the sine positional embeddings, the valid ratios, the encoder's reference grid - was
rebuilt on every forward: ~70 sub-10-us launches per micro-batch.  ``memo_on(t, tag, build)``
returns the value built for the SAME tensor object at the SAME version counter, and rebuilds
otherwise; entries hold a reference to their key tensor, so an ``id`` can never be recycled while its
entry lives.  Results are shared between calls: callers must not write into them (none does).
Disabled while autograd is recording.
"""
import torch

_MEMO = {}
_MAX = 32          # entries; the largest values are positional embeddings (~4 MB per frame)


def memo_on(t, tag, build):
    if torch.is_grad_enabled() or not isinstance(t, torch.Tensor):
        return build()
    key = (id(t), tag)
    hit = _MEMO.get(key)
    if hit is not None and hit[0] is t and hit[1] == t._version:
        _MEMO[key] = _MEMO.pop(key)             # most recently used last
        return hit[2]
    val = build()
    _MEMO.pop(key, None)
    while len(_MEMO) >= _MAX:                   # evict the least recently used entry only: entries keyed on per-call
        _MEMO.pop(next(iter(_MEMO)))            # tensors (valid ratios of one forward) must not flush the mask's
    _MEMO[key] = (t, t._version, val)
    return val


def clear():
    _MEMO.clear()
