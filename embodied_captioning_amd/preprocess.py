"""Object crops resized on the device, bit-exact with the reference's host-side PIL path.

The reference expands every detected box by 20 %, crops it out of the BGR frame, swaps to RGB and hands a PIL image to
the captioner, whose HF processor resizes it with ``Image.resize((S, S), resample=BICUBIC)``
(reference ``detector/pseudolabeler.py:629-643,670-675``; ``captioner/models/blip/blip.py`` processor call).  Here the
crop rectangles stay on the host (integer arithmetic), the integer filter tables of Pillow's resample are built on the
host in doubles (O(S) per box, `pil_bicubic_coeffs` - a restatement of Pillow's ``Resample.c::precompute_coeffs`` +
``normalize_coeffs_8bpc``), and every per-pixel operation runs in ``cap_crop_resize_u8`` (csrc/preprocess.hip).

No CPU fallback: without the HIP library / a GPU this module raises.
"""
from __future__ import annotations

import ctypes as C
from functools import lru_cache
from typing import Sequence, Tuple

import numpy as np
import torch

from . import _native as N

PRECISION_BITS = 32 - 8 - 2


@lru_cache(maxsize=4096)
def pil_bicubic_coeffs(in_size: int, out_size: int) -> Tuple[np.ndarray, np.ndarray]:
    """Pillow's bicubic tables for resizing `in_size` samples to `out_size` (whole-image box):
    bounds int32 [out, 2] = (first input index, tap count), coeffs int32 [out, ksize] (22-bit fixed point).
    Same double-precision operations in the same order as Resample.c, vectorised over the output index."""
    if in_size < 1 or out_size < 1:
        raise ValueError(f"resize {in_size} -> {out_size}")
    scale = in_size / out_size
    filterscale = scale if scale > 1.0 else 1.0
    support = 2.0 * filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    inv = 1.0 / filterscale
    center = (np.arange(out_size, dtype=np.float64) + 0.5) * scale
    xmin = np.maximum((center - support + 0.5).astype(np.int64), 0)          # C (int) cast: truncation, values are > -1
    xmax = np.minimum((center + support + 0.5).astype(np.int64), in_size) - xmin
    taps = np.arange(ksize, dtype=np.float64)[None, :]
    x = np.abs((taps + xmin[:, None] - center[:, None] + 0.5) * inv)
    a = -0.5
    w = np.where(x < 1.0, ((a + 2.0) * x - (a + 3.0)) * x * x + 1.0, np.where(x < 2.0, (((x - 5.0) * x + 8.0) * x - 4.0) * a, 0.0))
    valid = np.arange(ksize)[None, :] < xmax[:, None]
    w = np.where(valid, w, 0.0)
    ww = np.zeros(out_size, dtype=np.float64)
    for t in range(ksize):                    # Pillow sums the weights in tap order; padded taps add exactly 0.0
        ww = ww + w[:, t]
    w = np.where((ww != 0.0)[:, None], w / np.where(ww != 0.0, ww, 1.0)[:, None], w)
    fixed = np.where(w < 0, -0.5 + w * (1 << PRECISION_BITS), 0.5 + w * (1 << PRECISION_BITS)).astype(np.int64)   # (int): toward zero
    fixed = np.where(valid, fixed, 0).astype(np.int32)
    bounds = np.stack([xmin, xmax], axis=1).astype(np.int32)
    bounds.setflags(write=False); fixed.setflags(write=False)
    return bounds, fixed


def shorter_side_geometry(w: int, h: int, size: int) -> Tuple[int, int, int, int]:
    """open_clip's inference transform as the CoCa plugin applies it (captioner/models/coca/coca.py::preprocess): bicubic
    resize of the shorter side to `size`, then a centre crop.  -> (resized width, resized height, left, top)."""
    scale = size / min(w, h)
    nw, nh = max(size, round(w * scale)), max(size, round(h * scale))
    return nw, nh, (nw - size) // 2, (nh - size) // 2


def crop_resize_u8(frame, rects: Sequence[Sequence[int]], size: int, bgr: bool = False,
                   device: str | torch.device = "cuda:0", center_crop: bool = False) -> torch.Tensor:
    """frame uint8 [H, W, 3] (numpy or torch, host or device) + integer rectangles (x1, y1, x2, y2) (parts outside the
    frame read as zeros, as Image.crop pads) -> uint8 [n, size, size, 3] RGB on the device, equal to
    ``Image.fromarray(rgb).crop(r).resize((size, size), BICUBIC)`` for every rectangle; with `center_crop` to the
    aspect-preserving form ``resize(shorter side -> size)`` + centre crop (`shorter_side_geometry`): only the kept
    size x size window of the resized crop is computed - its rows of the two filter tables."""
    if not torch.cuda.is_available():
        raise N.CaptionerHipError("crop_resize_u8 needs a GPU; there is no CPU fallback in the product path")
    lib = N.load_library()
    dev = torch.device(device)
    if isinstance(frame, np.ndarray):
        frame = torch.from_numpy(np.ascontiguousarray(frame))
    if frame.dtype != torch.uint8 or frame.dim() != 3 or frame.shape[2] != 3:
        raise ValueError(f"frame must be uint8 [H, W, 3], got {frame.dtype} {tuple(frame.shape)}")
    H, W = int(frame.shape[0]), int(frame.shape[1])
    rects = np.asarray(rects, dtype=np.int64).reshape(-1, 4)
    n = rects.shape[0]
    if n == 0:
        return torch.empty((0, size, size, 3), dtype=torch.uint8, device=dev)
    x1, y1, x2, y2 = rects.T
    if (x2 <= x1).any() or (y2 <= y1).any() or (np.abs(rects) > 1 << 20).any():
        raise ValueError(f"empty or absurd crop rectangle: {rects.tolist()}")
    # a rectangle may leave the frame: Image.crop pads with zeros and so does the kernel (the reference's expand_box clamps
    # x to the frame height and y to its width, so this happens on non-square frames)
    if center_crop:
        tabs_h, tabs_v = [], []
        for w, h in zip((x2 - x1).tolist(), (y2 - y1).tolist()):
            nw, nh, left, top = shorter_side_geometry(int(w), int(h), size)
            bh, kh = pil_bicubic_coeffs(int(w), nw)
            bv, kv = pil_bicubic_coeffs(int(h), nh)
            tabs_h.append((bh[left:left + size], kh[left:left + size]))
            tabs_v.append((bv[top:top + size], kv[top:top + size]))
    else:
        tabs_h = [pil_bicubic_coeffs(int(w), size) for w in (x2 - x1)]
        tabs_v = [pil_bicubic_coeffs(int(h), size) for h in (y2 - y1)]
    KH = max(t[1].shape[1] for t in tabs_h)
    KV = max(t[1].shape[1] for t in tabs_v)
    hb = np.stack([t[0] for t in tabs_h]); vb = np.stack([t[0] for t in tabs_v])
    hk = np.zeros((n, size, KH), dtype=np.int32); vk = np.zeros((n, size, KV), dtype=np.int32)
    for i in range(n):
        hk[i, :, :tabs_h[i][1].shape[1]] = tabs_h[i][1]
        vk[i, :, :tabs_v[i][1].shape[1]] = tabs_v[i][1]
    # one upload for all tables
    blob = np.concatenate([rects.astype(np.int32).ravel(), hb.ravel(), hk.ravel(), vb.ravel(), vk.ravel()])
    with torch.cuda.device(dev):
        frame_d = frame.to(dev, non_blocking=True).contiguous()
        tab = torch.from_numpy(blob).to(dev, non_blocking=True)
        out = torch.empty((n, size, size, 3), dtype=torch.uint8, device=dev)
        base, o = tab.data_ptr(), 0
        ptrs = []
        for cnt in (n * 4, hb.size, hk.size, vb.size, vk.size):
            ptrs.append(C.c_void_p(base + 4 * o)); o += cnt
        N.check(lib.cap_crop_resize_u8(C.c_void_p(frame_d.data_ptr()), H, W, int(bool(bgr)), ptrs[0], ptrs[1], ptrs[2], KH,
                                       ptrs[3], ptrs[4], KV, n, size, C.c_void_p(out.data_ptr()),
                                       C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "cap_crop_resize_u8")
        out.record_stream(torch.cuda.current_stream(dev))
        tab.record_stream(torch.cuda.current_stream(dev)); frame_d.record_stream(torch.cuda.current_stream(dev))
    return out
