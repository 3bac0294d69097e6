"""Clip variant of util/misc.py (ref util/misc_multi.py:319-340): a clip arrives as ONE tensor
with its T frames stacked on the channel axis, [T*C, H, W]; ``split=True`` cuts it back into T
frames of ``channel_size`` channels so the batch axis of the model input is the frame axis."""
from typing import List

from torch import Tensor

from .misc import (NestedTensor, _pad_to_common, get_rank, get_world_size, inverse_sigmoid,  # noqa: F401
                   is_dist_avail_and_initialized, is_main_process)


def nested_tensor_from_tensor_list(tensor_list: List[Tensor], split=True, channel_size=3) -> NestedTensor:
    frames = []
    for t in tensor_list:
        frames.extend(t.split(channel_size, dim=0) if split else [t])
    return _pad_to_common(frames)


def collate_fn(batch, use_depth=False):
    """DataLoader collate of (clip, target) samples; clips are cut into frames of 4 (RGB-D) or 3 channels
    (ref util/misc_multi.py:304-308)."""
    columns = list(zip(*batch))
    columns[0] = nested_tensor_from_tensor_list(columns[0], split=True, channel_size=4 if use_depth else 3)
    return tuple(columns)
