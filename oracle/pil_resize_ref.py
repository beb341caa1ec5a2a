"""CPU restatement of Pillow's bicubic resample for 8-bit RGB images.  TEST INFRASTRUCTURE - NOT PRODUCT CODE.

The reference resizes every object crop with the HF image processor, i.e. `PIL.Image.resize((S, S), resample=BICUBIC)`
(reference `captioner/models/blip/blip.py` -> `BlipImageProcessor.resize`; crops made by `detector/pseudolabeler.py:
670-675`).  Pillow is a third-party dependency (present in this image), so the algorithm is restated from its published
source (`src/libImaging/Resample.c`: `precompute_coeffs`, `normalize_coeffs_8bpc`, `ImagingResampleHorizontal_8bpc`,
`ImagingResampleVertical_8bpc`) and PINNED by running Pillow itself on the same inputs (tests/test_preprocess_cpu.py).

Two separable passes, horizontal first, each output value = clip8((2^21 + sum_k pixel_k * coeff_k) >> 22) with integer
coefficients round_half_away(w * 2^22); the horizontal result is rounded to uint8 before the vertical pass reads it.
"""
from __future__ import annotations

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def _bicubic(x: np.ndarray) -> np.ndarray:
    a = -0.5
    x = np.abs(x)
    near = ((a + 2.0) * x - (a + 3.0)) * x * x + 1.0
    far = (((x - 5.0) * x + 8.0) * x - 4.0) * a
    return np.where(x < 1.0, near, np.where(x < 2.0, far, 0.0))


def coeffs(in_size: int, out_size: int):
    """-> (bounds int32 [out, 2] = (first input index, tap count), k int32 [out, ksize]) for the whole-image box."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = _bicubic((np.arange(xmax, dtype=np.float64) + xmin - center + 0.5) * ss)
        ww = 0.0
        for v in w:                       # Pillow accumulates in tap order
            ww += float(v)
        if ww != 0.0:
            w = w / ww
        kk[xx, :xmax] = np.where(w < 0, (-0.5 + w * (1 << PRECISION_BITS)).astype(np.int64),
                                 (0.5 + w * (1 << PRECISION_BITS)).astype(np.int64))   # C (int) cast truncates toward zero
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _clip8(v: np.ndarray) -> np.ndarray:
    return np.clip(v >> PRECISION_BITS, 0, 255).astype(np.uint8)


def resize_bicubic(img: np.ndarray, out_w: int, out_h: int) -> np.ndarray:
    """img uint8 [H, W, C] -> uint8 [out_h, out_w, C], as Image.fromarray(img).resize((out_w, out_h), Image.BICUBIC)."""
    H, W, C = img.shape
    cur = img
    if out_w != W:
        b, k = coeffs(W, out_w)
        out = np.empty((H, out_w, C), dtype=np.uint8)
        src = cur.astype(np.int64)
        for xx in range(out_w):
            x0, n = b[xx]
            acc = (1 << (PRECISION_BITS - 1)) + np.tensordot(src[:, x0:x0 + n, :], k[xx, :n].astype(np.int64), axes=([1], [0]))
            out[:, xx, :] = _clip8(acc)
        cur = out
    if out_h != H:
        b, k = coeffs(H, out_h)
        out = np.empty((out_h, cur.shape[1], C), dtype=np.uint8)
        src = cur.astype(np.int64)
        for yy in range(out_h):
            y0, n = b[yy]
            acc = (1 << (PRECISION_BITS - 1)) + np.tensordot(k[yy, :n].astype(np.int64), src[y0:y0 + n], axes=([0], [0]))
            out[yy] = _clip8(acc)
        cur = out
    return cur
