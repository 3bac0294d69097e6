"""Input pipeline step in front of the path: resize to short side 600 /
long side <= 1333, /255, normalise, zero-pad the clip's frames to one size and build the padding mask -
fused into one HIP kernel per image (csrc/preprocess.hip) that reproduces Pillow's bilinear resampler
bit for bit.  Reference: /root/reference/inference.py:285-350 (``ResizeWithMax`` / ``resize``),
:422-450 (ToTensor + Normalize stacks; RGB ImageNet statistics, depth 0.48 / 0.28),
util/misc.py:338-356 (padding collate).

The host computes the resampling taps from the two sizes (Pillow's precompute_coeffs /
normalize_coeffs_8bpc formulas) once per size pair and keeps them on the device.
"""
import math

import numpy as np
import torch

from dfx import _lib
from util.misc import NestedTensor

RGB_MEAN, RGB_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
DEPTH_MEAN, DEPTH_STD = (0.48,), (0.28,)
_PRECISION_BITS = 32 - 8 - 2


def get_size_with_aspect_ratio(image_size, size, max_size=None):
    """(w, h), target short side, cap on the long side -> (oh, ow)."""
    w, h = image_size
    if max_size is not None:
        lo, hi = float(min(w, h)), float(max(w, h))
        if hi / lo * size > max_size:
            size = int(round(max_size * lo / hi))
    if (w <= h and w == size) or (h <= w and h == size):
        return h, w
    if w < h:
        return int(size * h / w), size
    return size, int(size * w / h)


def resample_taps(in_size, out_size):
    """Bilinear taps of one axis: bounds int32 [out,2] (first index, count), weights int32 [out,ksize]."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.float64)
    inv = 1.0 / filterscale
    for i in range(out_size):
        center = (i + 0.5) * scale
        lo = max(int(center - support + 0.5), 0)
        n = min(int(center + support + 0.5), in_size) - lo
        w = np.maximum(0.0, 1.0 - np.abs((np.arange(n) + lo - center + 0.5) * inv))
        total = w.sum()
        kk[i, :n] = w / total if total != 0.0 else w
        bounds[i] = (lo, n)
    one = float(1 << _PRECISION_BITS)
    fixed = np.where(kk < 0, np.trunc(-0.5 + kk * one), np.trunc(0.5 + kk * one)).astype(np.int32)
    return bounds, fixed


class ClipPreprocessor:
    def __init__(self, size=600, max_size=1333, device="cuda"):
        self.size, self.max_size, self.device = size, max_size, torch.device(device)
        self._taps = {}
        self._stats = {}

    def _axis(self, n_in, n_out):
        if n_in == n_out:
            return None, None, 0
        key = (n_in, n_out)
        if key not in self._taps:
            b, k = resample_taps(n_in, n_out)
            self._taps[key] = (torch.from_numpy(b).to(self.device), torch.from_numpy(k).to(self.device), k.shape[1])
        return self._taps[key]

    def _stat(self, values):
        if values not in self._stats:
            self._stats[values] = torch.tensor(values, dtype=torch.float32, device=self.device)
        return self._stats[values]

    def _run(self, img, oh, ow, mean, std, dst, mask):
        lib = _lib.load()
        if not (img.is_cuda and img.dtype == torch.uint8 and img.is_contiguous()):
            raise RuntimeError("preprocess: uint8 contiguous CUDA image [H,W,C] expected (no CPU path)")
        hs, ws, cs = img.shape
        xb, xk, kx = self._axis(ws, ow)
        yb, yk, ky = self._axis(hs, oh)
        hp, wp = dst.shape[-2:]
        ptr = lambda t: 0 if t is None else t.data_ptr()  # noqa: E731
        with torch.cuda.device(img.device):
            rc = lib.dfx_preprocess_u8_f32(img.data_ptr(), hs, ws, cs, ptr(xb), ptr(xk), kx, ptr(yb), ptr(yk), ky, oh, ow,
                                           self._stat(mean).data_ptr(), self._stat(std).data_ptr(), dst.data_ptr(),
                                           hp * wp, hp, wp, ptr(mask), torch.cuda.current_stream(img.device).cuda_stream)
        _lib.check(rc, "preprocess")

    def __call__(self, rgb_frames, depth_frames=None):
        """rgb_frames: list of uint8 [H,W,3] CUDA tensors; depth_frames: list of uint8 [H,W] / [H,W,1] or None.
        -> NestedTensor([T,3|4,Hp,Wp] float32, mask [T,Hp,Wp] bool), frames padded to the largest size."""
        sizes = [get_size_with_aspect_ratio((f.shape[1], f.shape[0]), self.size, self.max_size) for f in rgb_frames]
        hp, wp = max(s[0] for s in sizes), max(s[1] for s in sizes)
        T, C = len(rgb_frames), 3 if depth_frames is None else 4
        batch = torch.empty((T, C, hp, wp), dtype=torch.float32, device=self.device)
        mask = torch.empty((T, hp, wp), dtype=torch.uint8, device=self.device)
        for t, (rgb, (oh, ow)) in enumerate(zip(rgb_frames, sizes)):
            self._run(rgb, oh, ow, RGB_MEAN, RGB_STD, batch[t, :3], mask[t])
            if depth_frames is not None:
                d = depth_frames[t]
                d = d.unsqueeze(-1) if d.dim() == 2 else d
                if tuple(d.shape[:2]) != tuple(rgb.shape[:2]):
                    raise RuntimeError("depth and RGB of a frame must have the same size")
                self._run(d.contiguous(), oh, ow, DEPTH_MEAN, DEPTH_STD, batch[t, 3:4], None)
        return NestedTensor(batch, mask.bool())
