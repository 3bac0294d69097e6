"""The slice of the reference's util/misc.py that sits on the inference path:
``NestedTensor`` / ``nested_tensor_from_tensor_list`` (ref util/misc.py:338-383) and
``inverse_sigmoid`` (ref :531-535), plus the process-rank helpers the model builders read.
Training bookkeeping (MetricLogger, reduce_dict, ...) is out of scope.
"""
from typing import List, Optional

import torch
import torch.distributed as dist
from torch import Tensor


class NestedTensor(object):
    """A batch of images padded to one size plus the bool mask of the padding (True = pad)."""

    def __init__(self, tensors, mask: Optional[Tensor]):
        self.tensors = tensors
        self.mask = mask

    def to(self, device, non_blocking=False):
        mask = None if self.mask is None else self.mask.to(device, non_blocking=non_blocking)
        return NestedTensor(self.tensors.to(device, non_blocking=non_blocking), mask)

    def record_stream(self, *args, **kwargs):
        self.tensors.record_stream(*args, **kwargs)
        if self.mask is not None:
            self.mask.record_stream(*args, **kwargs)

    def decompose(self):
        return self.tensors, self.mask

    def __repr__(self):
        return str(self.tensors)


def _pad_to_common(images: List[Tensor]) -> NestedTensor:
    if images[0].ndim != 3:
        raise ValueError("not supported")
    c = max(im.shape[0] for im in images)
    h = max(im.shape[1] for im in images)
    w = max(im.shape[2] for im in images)
    first = images[0]
    batch = torch.zeros((len(images), c, h, w), dtype=first.dtype, device=first.device)
    mask = torch.ones((len(images), h, w), dtype=torch.bool, device=first.device)
    for i, im in enumerate(images):
        batch[i, : im.shape[0], : im.shape[1], : im.shape[2]] = im
        mask[i, : im.shape[1], : im.shape[2]] = False
    return NestedTensor(batch, mask)


def nested_tensor_from_tensor_list(tensor_list: List[Tensor]) -> NestedTensor:
    """Zero-pad [C,H,W] images to the largest H and W; mask marks the padding."""
    return _pad_to_common(list(tensor_list))


def collate_fn(batch):
    """DataLoader collate of (image, target) samples: images padded into one NestedTensor, the rest transposed
    (ref util/misc.py:304-307)."""
    columns = list(zip(*batch))
    columns[0] = nested_tensor_from_tensor_list(columns[0])
    return tuple(columns)


def inverse_sigmoid(x, eps=1e-5):
    x = x.clamp(min=0, max=1)
    return torch.log(x.clamp(min=eps) / (1 - x).clamp(min=eps))


def is_dist_avail_and_initialized():
    return dist.is_available() and dist.is_initialized()


def get_world_size():
    return dist.get_world_size() if is_dist_avail_and_initialized() else 1


def get_rank():
    return dist.get_rank() if is_dist_avail_and_initialized() else 0


def is_main_process():
    return get_rank() == 0
