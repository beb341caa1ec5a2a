#!/usr/bin/env python
"""Box crops of a dataloader batch of frames (the pseudo-labeler: 1280 x 1280 frames, a few boxes each): one `crop_resize_u8` call per
frame (whole frame uploaded) against `crop_resize_u8_frames` (only the boxes' pixels, one packed upload, two launches).
    python tools/box_frames_bench.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from embodied_captioning_amd.preprocess import crop_resize_u8, crop_resize_u8_frames  # noqa: E402

rng = np.random.default_rng(0)
F, S = 64, 224
frames = [rng.integers(0, 256, size=(1280, 1280, 3), dtype=np.uint8) for _ in range(F)]
rects = []
for _ in range(F):
    rs = []
    for _ in range(3):
        w, h = int(rng.integers(80, 500)), int(rng.integers(80, 500))
        x, y = int(rng.integers(0, 1280 - w)), int(rng.integers(0, 1280 - h))
        rs.append((x, y, x + w, y + h))
    rects.append(rs)


def per_frame():
    return torch.cat([crop_resize_u8(f, r, S, bgr=True) for f, r in zip(frames, rects)])


def packed():
    return crop_resize_u8_frames(frames, rects, S, bgr=True)


res = {}
for name, fn in (("one call per frame (whole frame uploaded)", per_frame), ("all boxes of all frames, packed", packed)):
    out = fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        out = fn()
    torch.cuda.synchronize()
    res[name] = (out, (time.perf_counter() - t0) / 3)
a, b = res.values()
print(f"{F} frames 1280x1280, {3 * F} boxes -> {S}x{S}: " + "; ".join(f"{k}: {1e3 * v[1]:.1f} ms" for k, v in res.items()) + f"; equal: {torch.equal(a[0], b[0])}")
