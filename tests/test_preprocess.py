"""Preprocessing.  CPU: the numpy restatement of Pillow's resampler against the
installed Pillow (the library the reference calls through torchvision) and the size rule; GPU: the
fused HIP kernel against both, bit for bit."""
import numpy as np
import pytest
import torch

SHAPES = [(773, 1024, 3), (480, 640, 3), (100, 37, 1), (600, 800, 3), (50, 50, 3), (333, 1000, 1), (1080, 1920, 3)]


def _pil_resize(img, oh, ow):
    from PIL import Image
    pil = Image.fromarray(img if img.shape[2] == 3 else img[:, :, 0])
    return np.asarray(pil.resize((ow, oh), Image.BILINEAR)).reshape(oh, ow, img.shape[2])


@pytest.mark.parametrize("shape", SHAPES)
def test_oracle_resampler_equals_pillow(shape):
    from oracle import preprocess_oracle as po
    img = np.random.default_rng(sum(shape)).integers(0, 256, size=shape, dtype=np.uint8)
    oh, ow = po.get_size_with_aspect_ratio((shape[1], shape[0]), 600, 1333)
    assert np.array_equal(po.resize_u8(img, oh, ow), _pil_resize(img, oh, ow))


def test_size_rule_and_host_taps_match_the_oracle():
    from models import preprocess as pp
    from oracle import preprocess_oracle as po
    for wh in [(1024, 773), (640, 480), (37, 100), (1920, 1080), (600, 600), (500, 3000)]:
        assert pp.get_size_with_aspect_ratio(wh, 600, 1333) == po.get_size_with_aspect_ratio(wh, 600, 1333)
    assert pp.get_size_with_aspect_ratio((1024, 773), 600, 1333) == (600, 794)       # the sample image of config A
    for a, b in [(1024, 794), (773, 600), (480, 600), (37, 370)]:
        ba, ka = pp.resample_taps(a, b)
        bo, ko = po.coeffs(a, b)
        assert np.array_equal(ba, bo) and np.array_equal(ka, ko)


@pytest.mark.gpu
def test_fused_kernel_equals_pillow_pipeline():
    from models.preprocess import DEPTH_MEAN, DEPTH_STD, RGB_MEAN, RGB_STD, ClipPreprocessor
    from oracle import preprocess_oracle as po
    rng = np.random.default_rng(5)
    sizes = [(773, 1024), (480, 640), (600, 800), (300, 900)]
    rgbs = [rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8) for h, w in sizes]
    deps = [rng.integers(0, 256, size=(h, w), dtype=np.uint8) for h, w in sizes]
    pre = ClipPreprocessor(600, 1333)
    nt = pre([torch.from_numpy(a).cuda() for a in rgbs], [torch.from_numpy(a).cuda() for a in deps])
    out, mask = nt.tensors.cpu().numpy(), nt.mask.cpu().numpy()
    hp, wp = out.shape[-2:]
    for t, (rgb, dep) in enumerate(zip(rgbs, deps)):
        want_rgb = po.preprocess(rgb, RGB_MEAN, RGB_STD, 600, 1333)
        want_d = po.preprocess(dep[:, :, None], DEPTH_MEAN, DEPTH_STD, 600, 1333)
        oh, ow = want_rgb.shape[1:]
        # Pillow end to end as well (resize) + the float steps of ToTensor / Normalize
        pil = _pil_resize(rgb, oh, ow).astype(np.float32) / np.float32(255)
        pil = ((pil - np.asarray(RGB_MEAN, np.float32)) / np.asarray(RGB_STD, np.float32)).transpose(2, 0, 1)
        assert np.array_equal(want_rgb, pil)
        assert np.array_equal(out[t, :3, :oh, :ow], want_rgb), t
        assert np.array_equal(out[t, 3:, :oh, :ow], want_d), t
        assert not mask[t, :oh, :ow].any() and mask[t, oh:, :].all() and mask[t, :, ow:].all()
        assert (out[t, :, oh:, :] == 0).all() and (out[t, :, :, ow:] == 0).all()
    assert (hp, wp) == (max(po.get_size_with_aspect_ratio((w, h), 600, 1333)[0] for h, w in sizes),
                        max(po.get_size_with_aspect_ratio((w, h), 600, 1333)[1] for h, w in sizes))
