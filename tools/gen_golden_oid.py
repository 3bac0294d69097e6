"""Pixels of the reference's sample image of BASELINE.json configs[0] as a data fixture (build container only):
    python tools/gen_golden_oid.py   ->   tests/golden/oid_sample.npz
/root/reference/sample_dataset/OID/images/0000b7e1500c94d7.jpg (773 x 1024 RGB) and its depth map
sample_dataset/OID/depth/0000b7e1500c94d7.jpg (8-bit grey), decoded here with Pillow - the decoder the reference's
Image.open() uses (inference.py:660-716) - and stored as uint8 arrays.  The GPU box has no /root/reference."""
import os

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = "/root/reference/sample_dataset/OID"
NAME = "0000b7e1500c94d7.jpg"
rgb = np.asarray(Image.open(os.path.join(SRC, "images", NAME)).convert("RGB"))
depth = np.asarray(Image.open(os.path.join(SRC, "depth", NAME)))
assert rgb.shape == (1024, 773, 3) and depth.shape == (1024, 773) and rgb.dtype == depth.dtype == np.uint8
out = os.path.join(ROOT, "tests", "golden", "oid_sample.npz")
np.savez_compressed(out, rgb=rgb, depth=depth)
print("wrote", out, f"{os.path.getsize(out) / 1e6:.2f} MB")
