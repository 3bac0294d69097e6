"""CPU restatement of the reference's image preprocessing -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

The reference resizes PIL images with ``torchvision.transforms.functional.resize`` (bilinear, i.e.
``PIL.Image.resize(size, BILINEAR)``), converts with ``ToTensor`` (/255) and normalises
(/root/reference/inference.py:285-350 ``ResizeWithMax`` / ``resize``, :422-450 the transform stacks;
RGB mean/std ImageNet, depth 0.48 / 0.28).  Pillow is third party: its resampler
(src/libImaging/Resample.c: precompute_coeffs, normalize_coeffs_8bpc, ImagingResampleHorizontal_8bpc /
Vertical_8bpc; 22-bit fixed point, horizontal pass first, uint8 intermediate) is restated here in numpy
and PINNED against the installed Pillow by tests/test_preprocess.py.
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def get_size_with_aspect_ratio(image_size, size, max_size=None):
    """(w, h) of the source, target short side, cap on the long side -> (oh, ow)   (inference.py:321-340)"""
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


def coeffs(in_size, out_size):
    """Pillow's bilinear resampling taps for one axis: bounds [out,2] = (first input index, count) and
    fixed-point weights [out, ksize] (int32)."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.float64)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        x = np.arange(xmax)
        w = np.maximum(0.0, 1.0 - np.abs((x + xmin - center + 0.5) * ss))
        ww = w.sum()
        if ww != 0.0:
            w = w / ww
        kk[xx, :xmax] = w
        bounds[xx] = (xmin, xmax)
    fixed = np.where(kk < 0, np.trunc(-0.5 + kk * (1 << PRECISION_BITS)), np.trunc(0.5 + kk * (1 << PRECISION_BITS)))
    return bounds, fixed.astype(np.int32)


def _pass(img, bounds, kk, axis):
    """One resampling pass over ``axis`` of a uint8 array [H,W,C]."""
    img = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((bounds.shape[0],) + img.shape[1:], dtype=np.uint8)
    for i in range(bounds.shape[0]):
        lo, n = bounds[i]
        acc = (1 << (PRECISION_BITS - 1)) + np.tensordot(kk[i, :n].astype(np.int64), img[lo:lo + n], axes=(0, 0))
        out[i] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def resize_u8(img, oh, ow):
    """img uint8 [H,W,C] -> uint8 [oh,ow,C], Pillow BILINEAR semantics (horizontal pass, then vertical)."""
    h, w = img.shape[:2]
    out = img
    if ow != w:
        out = _pass(out, *coeffs(w, ow), axis=1)
    if oh != h:
        out = _pass(out, *coeffs(h, oh), axis=0)
    return out


def preprocess(img, mean, std, size=600, max_size=1000):
    """uint8 [H,W,C] -> float32 [C,oh,ow]: resize, /255, (x - mean) / std, all in float32 like
    ToTensor + Normalize."""
    oh, ow = get_size_with_aspect_ratio((img.shape[1], img.shape[0]), size, max_size)
    r = resize_u8(img, oh, ow).astype(np.float32) / np.float32(255)
    r = (r - np.asarray(mean, np.float32)) / np.asarray(std, np.float32)
    return np.ascontiguousarray(r.transpose(2, 0, 1))
