#!/usr/bin/env python3
"""Fixture generator: block means of the reference's own rendered image.

/root/reference/FinalRendered_Specular.jpg (512x512, 8-bit, lossy) is the only output of the real
OpenCL path tracer that the reference holds.  This script decodes it (PIL, a pure decoder) and
stores the mean R,G,B of every 32x32-pixel block (16x16 grid) and every 8x8-pixel block (64x64
grid) as float32 -- data only; the JPG itself is not copied.  Run in the build container:
    python tests/golden/make_jpg_blocks.py
"""
import os
import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
img = np.asarray(Image.open("/root/reference/FinalRendered_Specular.jpg").convert("RGB")).astype(np.float64)
assert img.shape == (512, 512, 3)
for g in (16, 64):
    blocks = img.reshape(g, 512 // g, g, 512 // g, 3).mean(axis=(1, 3)).astype(np.float32)
    np.save(os.path.join(HERE, "reference_jpg_blocks_%dx%d.npy" % (g, g)), blocks)
    print(g, blocks.shape, blocks.mean(axis=(0, 1)))
